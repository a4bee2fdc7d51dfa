#!/bin/bash
# CPU model of the accept loop with / without batch-accept (tools/batch_accept_model.c): trips per group on synthetic polygon and natural-image-like frames
set -e
cd "$(dirname "$0")/.."
N=${1:-3}
mkdir -p /tmp/bam
# the oracle with its region_grow renamed; the model's own region_grow (forward-declared in its place) serves refine as well
sed 's/^static void region_grow(lsd_t \*L, int sx, int sy, regpt \*reg, int \*reg_size, double \*reg_angle, double prec)$/static void region_grow(lsd_t *L, int sx, int sy, regpt *reg, int *reg_size, double *reg_angle, double prec);\nstatic void region_grow_ref(lsd_t *L, int sx, int sy, regpt *reg, int *reg_size, double *reg_angle, double prec)/' oracle/lsd_oracle.c > /tmp/bam/lsd_inc.c
cat /tmp/bam/lsd_inc.c tools/batch_accept_model.c > /tmp/bam/model.c
gcc -O2 -ffp-contract=off -Ioracle ${BAM_FLAGS} -o /tmp/bam/model /tmp/bam/model.c oracle/orb_oracle.c oracle/timing.c -lm 2>&1 | grep -v 'defined but not used' | grep -E 'error|undefined' || true
python - <<PY
import sys
sys.path.insert(0, ".")
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
for i in range($N):
    synth_frame(i).tofile("/tmp/bam/poly%d.raw" % i)
    natural_frame(i).tofile("/tmp/bam/nat%d.raw" % i)
PY
echo "== polygons"; /tmp/bam/model /tmp/bam/poly*.raw | tail -3
echo "== natural";  /tmp/bam/model /tmp/bam/nat*.raw | tail -3
