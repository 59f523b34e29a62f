#!/bin/bash
# profiles/r06_launch_anatomy.log: the per-launch fixed cost of the 128-row GEMMs, decomposed, on ONE box:
#   (b) wo / qkv / w1 / w2 at tp = 1 in situ, (a) the tp = 8 shard in loop-back (all-reduce and sequence-parallel forms), (c) fp8a,
#   and the 256-row kernel's wait cycles at num_images = 4.  Run on the GPU box from the repo root.
O=${1:-gpurun_out/r06_launch_anatomy.log}
export BD_HIP_LIB=$PWD/bitdance_amd/libbitdance_hip_stamp.so
{
  echo "##### (b) tp = 1, bf16, in situ"; timeout 200 python tools/launch_anatomy.py bf16 2>&1 | grep -v amdgpu.ids
  echo; echo "##### (a) tp = 8 shard, rank 0 in loop-back, all-reduce form"; timeout 200 python tools/launch_anatomy.py bf16 --tp-shard 0/8 2>&1 | grep -v amdgpu.ids
  echo; echo "##### (a') tp = 8 shard, rank 0 in loop-back, sequence-parallel form"; timeout 200 python tools/launch_anatomy.py bf16 --tp-shard 0/8 tp.seq=1 2>&1 | grep -v amdgpu.ids
  echo; echo "##### (c) tp = 1, fp8a"; timeout 200 python tools/launch_anatomy.py fp8a 2>&1 | grep -v amdgpu.ids
  echo; echo "##### num_images = 4: the 256-row kernel (wide:*) and the row kernels at 512 rows"; BD_ANATOMY_B=4 timeout 300 python tools/launch_anatomy.py bf16 6 2>&1 | grep -v amdgpu.ids
} > $O
