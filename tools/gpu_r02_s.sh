#!/bin/bash
# session S: correctness bisect of the bench-shape forward
cd "$(dirname "$0")/.." && export VD_QUIET=1
B=$PWD/versatile-diffusion_amd/build
D=/tmp/fd; mkdir -p $D
run() { env $2 timeout 300 python tools/fwd_dump.py $D/$1.pt 2>&1 | tail -1; }
run ref "VD_LN_FOLD=0 VD_FWD_OVERRIDE=0"
for c in 1 2 3 4 7 14 15 24; do
run o$c "VD_LN_FOLD=0 VD_FWD_OVERRIDE=$c"
python tools/fwd_dump.py --cmp $D/o$c.pt $D/ref.pt
done
