cd $GRAFT_REPO_ROOT
Q1="head.qkv.S=2,head.qkv.nw=2,head.qkv.kw=2,head.w1.S=2,head.w1.nw=2,head.w1.kw=2"
Q2="head.qkv.S=1,head.qkv.nw=2,head.qkv.kw=2,head.w1.S=1,head.w1.nw=2,head.w1.kw=2"
Q3="head.qkv.S=2,head.qkv.nw=4,head.qkv.kw=2,head.w1.S=2,head.w1.nw=4,head.w1.kw=2"
Q4="head.qkv.S=4,head.qkv.nw=2,head.qkv.kw=2,head.w1.S=2,head.w1.nw=2,head.w1.kw=2"
R1="head.wo.S=1,head.wo.nw=2,head.wo.kw=2,head.w2.S=1,head.w2.nw=2,head.w2.kw=1"
R2="head.wo.S=1,head.wo.nw=2,head.wo.kw=2,head.w2.S=2,head.w2.nw=2,head.w2.kw=1"
R3="head.wo.S=2,head.wo.nw=2,head.wo.kw=2,head.w2.S=3,head.w2.nw=2,head.w2.kw=1"
python tools/head_sweep.py 3 50 "tp.seq=1;tp.seq=1,$Q1;tp.seq=1,$Q2;tp.seq=1,$Q3;tp.seq=1,$Q4;tp.seq=1,$Q1,$R1;tp.seq=1,$Q1,$R2;tp.seq=1,$Q1,$R3;tp.seq=0,$Q1,$R1;tp.seq=1,$Q1,$R1,sp_wait=0" bf16 --tp-shard 0/8 --loopback
