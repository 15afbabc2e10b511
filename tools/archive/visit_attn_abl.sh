# phase ablations of the pipelined long-sequence attention kernel (CCDM_EXPERIMENTS build, results wrong by construction): where does the time go?
set -u
export TMPDIR=/tmp
export CCDM_LIB=$PWD/tools/ab/exp.so
for a in 0 1 2 3 4 8 12 16 32 48 15 51 60; do
  echo -n "ABL=$a  "; CCDM_ATTN_SPLIT_MODE=6 CCDM_ATTN_ABL=$a timeout 300 python tools/bench_attention.py 2>&1 | grep "T=8192"
done
