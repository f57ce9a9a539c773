cd ${GRAFT_REPO_ROOT:-/root/repo}
export HOST_PARITY_VERBOSE=2
bad=0
for i in $(seq 1 160); do
  timeout 100 tests/cpp/_build/host_parity > /tmp/hp.log 2>&1
  if ! grep -q "rel err 2.98e-08  device sum 113.153770071097 oracle sum 113.153770130350" /tmp/hp.log; then
    bad=$((bad+1)); echo "=== run $i"; grep "step \|discrete weights\|FAIL" /tmp/hp.log | cut -c1-150
  fi
done
echo "bad runs: $bad of 160"
