cd $GRAFT_REPO_ROOT
export BD_SWEEP_B=4
python tools/head_sweep.py 2 50 "ada_group=2;head.qkv.S=1;head.w1.S=1;head.qkv.S=1,head.w1.S=1;head.qkv.S=1,head.w1.S=1,wide.xcd=1;head.qkv.S=1,head.w1.S=1,ada_group=4;head.qkv.S=1,head.qkv.nw=4,head.w1.S=1,head.w1.nw=4" bf16
