cd $GRAFT_REPO_ROOT
R1="head.wo.S=1,head.wo.nw=2,head.wo.kw=2,head.w2.S=1,head.w2.nw=2,head.w2.kw=1"
B="tp.seq=1,$R1"
python tools/head_sweep.py 3 50 "$B,head.qkv.S=2,head.qkv.nw=4,head.qkv.kw=2,head.w1.S=2,head.w1.nw=4,head.w1.kw=2;$B,head.qkv.S=8,head.qkv.nw=4,head.qkv.kw=2,head.w1.S=2,head.w1.nw=4,head.w1.kw=2;$B,head.qkv.S=8,head.qkv.nw=4,head.qkv.kw=2,head.w1.S=8,head.w1.nw=4,head.w1.kw=2,w1_fused=0;$B,head.qkv.S=4,head.qkv.nw=2,head.qkv.kw=2,head.w1.S=4,head.w1.nw=2,head.w1.kw=2,w1_fused=0;$B,head.qkv.S=16,head.qkv.nw=8,head.qkv.kw=2,head.w1.S=16,head.w1.nw=8,head.w1.kw=2,w1_fused=0;$B,head.qkv.S=5,head.qkv.nw=4,head.qkv.kw=2,head.w1.S=5,head.w1.nw=4,head.w1.kw=2,w1_fused=0" bf16 --tp-shard 0/8 --loopback
