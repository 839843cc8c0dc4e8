cd tools
for c in ED25519 ED448 NIST256; do lg=20; [ $c = ED448 ] && lg=19; for t in u1 b1; do ./ecn_exp_${t}_$c.bin $lg 3; done; done
