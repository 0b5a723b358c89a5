#!/bin/bash
# Rehearsal of the driver's multi-GPU bench command on a ONE-GPU box (not a measurement): the same torchrun launch, N ranks
# sharing GPU 0, torch.distributed over gloo, glu_dist_* over the file transport of tests/cpp/mock_rccl.cpp.
#   bash tools/rehearse_multi_gpu.sh [ranks] [log2 pairs per rank]
# PLAIN=1: the plain command `python bench.py --gpus N ...` instead (bench.py starts torchrun itself as a child process).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-2}
L=${2:-22}
export GLU_MOCK_RCCL_DIR=$(mktemp -d /tmp/mock_rccl.XXXXXX)
export GLU_HIP_RCCL_LIB=$R/tests/cpp/bin/libmock_rccl.so
cd $R
if [ -n "$PLAIN" ]; then
    timeout 600 python bench.py --gpus $N --steps 4 --warmup 2 --log2-keys $L --rehearse-one-gpu
else
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
        bench.py --gpus $N --steps 4 --warmup 2 --log2-keys $L --rehearse-one-gpu
fi
rc=$?
rm -rf $GLU_MOCK_RCCL_DIR
exit $rc
