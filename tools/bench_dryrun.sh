# bench.py --gpus N end to end with ALL ranks on the one available device (gloo carries handles and flags): the protocol, not the figures
export SMARTIES_BENCH_PG=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/r06f
for n in 2 8; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700+n)) bench.py --gpus $n --steps 1000 --warmup 100 > gpurun_out/r06f/dry$n.log 2>&1
  echo "N=$n rc=$?"; grep "^{" gpurun_out/r06f/dry$n.log | tail -1 > gpurun_out/r06f/bench_gpus${n}_dryrun_one_device.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r06f/bench_gpus${n}_dryrun_one_device.json"))
    print("N=$n value %.0f ms/step %.4f scaling %s identical %s" % (d["value"], d["ms_per_step"], d["scaling"], d["config"].get("replicas_identical_after_timed_steps")), d["config"]["exchange_per_rank"][0][:60], d.get("weak_scaling_row",{}).get("ms_per_step"))
except Exception as e:
    print("N=$n FAILED", e); print(open("gpurun_out/r06f/dry$n.log").read()[-1500:])
PY
done
