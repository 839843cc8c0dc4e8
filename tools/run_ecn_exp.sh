cd tools
for t in v1 v1w3 v2 v2w3 v2w4; do timeout 120 ./ecn_exp_${t}_ED25519.bin 20 7; done
for c in ED448 NIST256; do lg=19; [ $c = ED448 ] && lg=18; for t in v1; do timeout 120 ./ecn_exp_${t}_$c.bin $lg 7; done; done
