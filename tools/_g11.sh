cd $GRAFT_REPO_ROOT
export BD_SWEEP_B=4
python tools/head_sweep.py 2 50 "ada_group=2;wide.xcd=1;wide.xcd=0;head.wo.S=3,head.w2.S=3;head.wo.S=4,head.w2.S=4;head.wo.S=7,head.w2.S=7;w1_fused=1;ada_group=4;ada_group=1;wide.xcd=1,head.wo.S=3,head.w2.S=3" bf16
