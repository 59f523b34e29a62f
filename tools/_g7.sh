cd $GRAFT_REPO_ROOT
for T in 8 4 2; do
python tools/head_sweep.py 3 50 "tp.seq=0,tp_shapes=0;tp.seq=0;tp.seq=1;tp.seq=1,sp_gsig=0;tp.seq=1,sp_wait=0;tp.seq=1,sp_inv=1" bf16 --tp-shard 0/$T --loopback
done
