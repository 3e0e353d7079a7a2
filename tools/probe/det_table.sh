#!/bin/bash
# Bit-reproducibility of graph-replayed training runs under host activity (tools/probe/det_probe4.py; small learning-test models).
# usage: bash tools/probe/det_table.sh [steps] [reps]
cd "$(dirname "$0")/../.."
S=${1:-60}; R=${2:-4}
P="python tools/probe/det_probe4.py bf16"
run() { echo "## $1"; shift; env "$@" 2>/dev/null | grep -v "^rep 0 readings" | tail -4; }
for K in superglue gluestick; do
echo "# ---- $K, $R repetitions of $S steps"
run "A: batches built on the CPU and uploaded (pageable) inside the loop, graph replay  [the learning test's loop]" GF_KIND=$K timeout 600 $P A $S $R
run "E: the same, host waits for every replay (= TrainStep(deterministic_replay=True))" GF_KIND=$K timeout 600 $P E $S $R
run "A, launched kernel by kernel (GF_EAGER=1)" GF_KIND=$K GF_EAGER=1 timeout 600 $P A $S $R
run "F: device batches made before the loop, nothing else on the host, graph replay" GF_KIND=$K timeout 600 $P F $S $R
run "F + a blocking pageable host-to-device copy (512 KB) per step (GF_SUB=h2d)" GF_KIND=$K GF_SUB=h2d timeout 600 $P F $S $R
run "F + an asynchronous pinned host-to-device copy per step (GF_SUB=h2d_pinned)" GF_KIND=$K GF_SUB=h2d_pinned timeout 600 $P F $S $R
run "F + a device-to-device copy per step (GF_SUB=d2d)" GF_KIND=$K GF_SUB=d2d timeout 600 $P F $S $R
run "F + an unrelated kernel per step (GF_SUB=kernel)" GF_KIND=$K GF_SUB=kernel timeout 600 $P F $S $R
run "F + random host sleeps in front of a step (GF_PERTURB=sleep)" GF_KIND=$K GF_PERTURB=sleep timeout 600 $P F $S $R
run "F + 16..224 CUs held by another stream for 1-3 ms under a step (GF_PERTURB=hold)" GF_KIND=$K GF_PERTURB=hold timeout 600 $P F $S $R
run "F + every CU's LDS filled with NaN patterns in front of every step (GF_PERTURB=dirtynan)" GF_KIND=$K GF_PERTURB=dirtynan timeout 600 $P F $S $R
done
