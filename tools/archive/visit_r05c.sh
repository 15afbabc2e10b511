# visit r05c: the interleaved commit (ILV) against HEAD: conv parity, single-layer A/B (old / new / new with the staged commit), block timeline, step A/B
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "conv or resblock or unet_step or trajectory" 2>&1 | tail -15
for rep in 1 2; do
CCDM_LIB=$PWD/tools/abx/old.so python tools/ab_conv.py 0,1,6,10
python tools/ab_conv.py 0,1,6,10
DBG=4096 python tools/ab_conv.py 0,1,6,10
done
CCDM_LIB=$PWD/tools/abx/abl.so TIMELINE=0 python tools/bench_conv.py 2>&1 | tail -2
sed -i 's#tools/ab/old.so#tools/abx/old.so#' tools/visit_ab.sh
bash tools/visit_ab.sh
