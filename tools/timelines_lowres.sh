export TMPDIR=/tmp
export CCDM_LIB=$PWD/tools/abx/abl.so
: > gpurun_out/timelines_r06.txt
for op in 5 6 10 12 13 65 66 71 72; do CCDM_TIMELINE_OP=$op timeout 300 python tools/timeline_op.py --no-pmc --substreams 1 --graph 0 2>/dev/null | tail -1 >> gpurun_out/timelines_r06.txt; done
for op in 15 16 17 20 24 25 26 29 39 40 52 53; do CCDM_TIMELINE_KS=1 CCDM_TIMELINE_OP=$op timeout 300 python tools/timeline_op.py --no-pmc --substreams 1 --graph 0 2>/dev/null | tail -1 >> gpurun_out/timelines_r06.txt; done
cut -c1-200 gpurun_out/timelines_r06.txt
