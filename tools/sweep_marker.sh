mkdir -p gpurun_out/r3i
for cfg in "0 4" "1 4" "2 4" "0 2" "0 0" "1 2"; do
  set -- $cfg
  N2M_MARKER_AT=$1 N2M_FWD_XCD_GROUP=$2 python bench.py --no-cpu-baseline > gpurun_out/r3i/b_m$1_g$2.json 2>/dev/null
  echo "marker=$1 group=$2: $(python tools/show_bench.py gpurun_out/r3i/b_m$1_g$2.json | grep -E 'samples/s|lookup|grid_encode_backward  |adam' | tr '\n' ' ' | cut -c1-420)"
done
