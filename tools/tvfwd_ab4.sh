mkdir -p gpurun_out/r6ab
export N2M_TV_FWD=1
bash tools/env_sweep.sh N2M_MARKER_AT "0 1 2" r6ab/marker 2>&1 | tee gpurun_out/r6ab/tvfwd_marker.txt
unset N2M_TV_FWD
bash tools/env_sweep.sh N2M_TV_FWD "0 1" r6ab/sdfcheck --recipe lego --diffuse 2>&1 | tee gpurun_out/r6ab/tvfwd_diffuse.txt
