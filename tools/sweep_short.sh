# distribution of 20-batch stream times against the pipeline's stream counts (device-level settings: one process each)
cd $GRAFT_REPO_ROOT
for cfg in "10 10" "8 8" "8 10" "6 8" "6 6" "10 6" "12 8"; do
  set -- $cfg
  MI_SNARK_VS_LANES=$1 MI_SNARK_VS_SLOTS=$2 timeout -k 10 120 python tools/short_stream_probe.py 20 "$1,$2" 2>/dev/null | grep "^lanes"
done
