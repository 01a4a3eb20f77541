mkdir -p gpurun_out/s6
for rep in 1 2; do
for q in default 2 1; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  DNM_TEST_NO_QUEUE_DEFAULT=1 DNM_FAKE_RCCL_STATS=1 timeout 500 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "native_schedule and (full-2 or full_partner-4)" --durations=5 > gpurun_out/s6/q_${q}_$rep.log 2>&1
  echo "== queues $q rep $rep"; grep "s call" gpurun_out/s6/q_${q}_$rep.log
done; done
