#!/bin/bash
# the DBM step's kernels in launch order (eager steps of tools/bench_configs-like loop): gpurun_out/<tag>/dbm_trace.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cat > /tmp/dbm_steps.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "video-mamba-suite_amd"))
from mamba_ssm.modules.mamba_new import Mamba as DBM
m = DBM(512, expand=1).cuda()
x = torch.randn(2, 2304, 512, device="cuda", dtype=torch.bfloat16, requires_grad=True)
g = torch.randn_like(x)
for _ in range(13):
    m.zero_grad(set_to_none=True); x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    y.backward(g)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o p --output-format csv -- python /tmp/dbm_steps.py > $O/prof.log 2>&1
python $R/tools/step_trace.py $O/prof/p_kernel_trace.csv 13 -v > $O/dbm_trace.txt
rm -rf $O/prof
cat $O/dbm_trace.txt | cut -c1-150
