set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "attention" 2>&1 | tail -8
echo "--- product"; timeout 300 python tools/bench_attention.py 2>&1 | grep "T=8192\|T=2048"
export CCDM_LIB=$PWD/tools/ab/exp.so
for a in 0; do echo -n "ABL=$a  "; CCDM_ATTN_SPLIT_MODE=6 CCDM_ATTN_ABL=$a timeout 300 python tools/bench_attention.py 2>&1 | grep "T=8192"; done
echo -n "mode5 "; CCDM_ATTN_SPLIT_MODE=5 timeout 300 python tools/bench_attention.py 2>&1 | grep "T=8192"
