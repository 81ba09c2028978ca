python -m pytest tests/test_gpu_torchsparse_ops.py -x -q -k "bf16_storage" > gpurun_out/e12_test.log 2>&1; echo "test rc=$?"; tail -4 gpurun_out/e12_test.log
