cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "half_tiles" 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400
BD_SWEEP_B=4 timeout 1200 python tools/head_sweep.py 3 6 "ada_group=4,half.form=0;ada_group=4,half.form=1;ada_group=4,half.form=0;ada_group=4,half.form=1" 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
export BD_HIP_LIB=$PWD/bitdance_amd/libbitdance_hip_stamp.so
for F in 1; do echo "##### half.form=$F"; BD_ANATOMY_B=4 timeout 300 python tools/launch_anatomy.py bf16 6 opt.half.form=$F 2>&1 | grep -v amdgpu.ids | grep -A3 "== wide:half:head.qkv" | cut -c1-330; done
