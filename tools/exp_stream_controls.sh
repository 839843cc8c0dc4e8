cd $GRAFT_REPO_ROOT
for P in X448 X25519; do
python tools/stream_controls.py $P
python tools/stream_controls.py $P
MA_FORCE_FAST=1 python tools/stream_controls.py $P
MA_FORCE_EXACT=1 python tools/stream_controls.py $P
for mb in 512 1024 2048 8192 32768; do MA_MAX_BLOCKS=$mb python tools/stream_controls.py $P; done
for pad in 32 512 8192 131072; do python tools/stream_controls.py $P $pad; done
python tools/stream_controls.py $P 0 22
python tools/stream_controls.py $P 0 23
done
