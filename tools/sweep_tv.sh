# where / at which priority the TV-terms kernel runs beside the step (engine.py: N2M_TV_AT, N2M_TV_PRIO); N2M_TV_SPLIT=0 = inside the fill
mkdir -p gpurun_out/r3k
for cfg in "0 0 0" "1 0 0" "1 0 1" "1 1 0" "1 2 0" "1 2 1"; do
  set -- $cfg
  N2M_TV_SPLIT=$1 N2M_TV_AT=$2 N2M_TV_PRIO=$3 python bench.py --no-cpu-baseline > gpurun_out/r3k/b_s$1_a$2_p$3.json 2>/dev/null
  echo "split=$1 at=$2 prio=$3: $(python tools/show_bench.py gpurun_out/r3k/b_s$1_a$2_p$3.json | grep -E 'samples/s|grid_encode_backward  |grad_total|composite|mlp_' | sed 's/  */ /g' | cut -d' ' -f1-12 | tr '\n' '|' | cut -c1-600)"
done
