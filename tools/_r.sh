python -m pytest tests/test_gpu_slab_live.py tests/test_gpu_native_comm.py -x -q -m gpu -s 2>&1 | grep -v "serach\|liqiud\|RCCL ver\|HIP ver\|ROCm ver\|Hostname\|Librccl\|amdgpu.ids" | tail -70
