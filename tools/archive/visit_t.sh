set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "${K:-trajectory or philox or sampler_bit or caller or softmax_output or c4_n16}" -s 2>&1 | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
