cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tp8 && timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_tp8 -o p -- python $GRAFT_REPO_ROOT/tools/head_sweep.py 1 50 "tp.seq=1" bf16 --tp-shard 0/8 --loopback
DB=$(ls /tmp/prof_tp8/*.db /tmp/prof_tp8/*/*.db 2>/dev/null | head -1)
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $DB gpurun_out/g3_kernel_stats_tp8_seq1.md | head -30
cd /tmp && rm -rf /tmp/prof_tp8b && timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_tp8b -o p -- python $GRAFT_REPO_ROOT/tools/head_sweep.py 1 50 "tp.seq=0" bf16 --tp-shard 0/8 --loopback
DB=$(ls /tmp/prof_tp8b/*.db /tmp/prof_tp8b/*/*.db 2>/dev/null | head -1)
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $DB gpurun_out/g3_kernel_stats_tp8_seq0.md | head -30
