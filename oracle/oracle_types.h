/* oracle/oracle_types.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * Type macros as the generators' header() emits them (pseudo.py:1394-1401, monty.py:1864-1871). */
#ifndef ORACLE_TYPES_H
#define ORACLE_TYPES_H
#include <stddef.h>
#include <stdint.h>
typedef uint64_t spint;
typedef int64_t sspint;
typedef unsigned __int128 dpint;
typedef __int128 sdpint;
#define ORACLE_CAT_(a, b) a##_##b
#define ORACLE_CAT(a, b) ORACLE_CAT_(a, b)
#define F(name) ORACLE_CAT(name, PRIME)
#endif
