cd $GRAFT_REPO_ROOT
C="head.qkv.S=2,head.qkv.nw=4,head.qkv.kw=2,head.w1.S=2,head.w1.nw=4,head.w1.kw=2"
R="head.wo.S=1,head.wo.nw=2,head.wo.kw=2,head.w2.S=1,head.w2.nw=2,head.w2.kw=2"
R2="head.wo.S=2,head.wo.nw=2,head.wo.kw=2,head.w2.S=2,head.w2.nw=2,head.w2.kw=2"
python tools/head_sweep.py 3 50 "tp.seq=1,sp_gsig=0;tp.seq=1,sp_gsig=0,$C;tp.seq=1,sp_gsig=0,$C,$R;tp.seq=1,sp_gsig=0,$C,$R2;tp.seq=0,$C,$R" bf16 --tp-shard 0/2 --loopback
python tools/head_sweep.py 3 50 "tp.seq=1,sp_gsig=0;tp.seq=1,sp_gsig=0,$R;tp.seq=1,sp_gsig=0,$R2;tp.seq=1,sp_gsig=0,head.qkv.S=2,head.w1.S=2;tp.seq=1,sp_gsig=0,head.qkv.S=2,head.w1.S=2,$R;tp.seq=0,$R" bf16 --tp-shard 0/4 --loopback
for T in 2 4; do
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$T && timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$T -o p -- python $GRAFT_REPO_ROOT/tools/head_sweep.py 1 50 "tp.seq=1,sp_gsig=0" bf16 --tp-shard 0/$T --loopback > /dev/null 2>&1
DB=$(ls /tmp/prof_$T/*.db /tmp/prof_$T/*/*.db 2>/dev/null | head -1)
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $DB gpurun_out/g9_kernel_stats_tp${T}_seq1.md | head -18
done
