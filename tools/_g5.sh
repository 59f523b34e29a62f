CFG="tp.seq=1,head.qkv.S=2,head.qkv.nw=4,head.qkv.kw=2,head.w1.S=2,head.w1.nw=4,head.w1.kw=2,head.wo.S=1,head.wo.nw=2,head.wo.kw=2,head.w2.S=1,head.w2.nw=2,head.w2.kw=1"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_a && timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_a -o p -- python $GRAFT_REPO_ROOT/tools/head_sweep.py 1 50 "$CFG" bf16 --tp-shard 0/8 --loopback > /dev/null 2>&1
DB=$(ls /tmp/prof_a/*.db /tmp/prof_a/*/*.db 2>/dev/null | head -1)
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $DB gpurun_out/g5_kernel_stats_tp8_seq1_cfgB.md | head -16
