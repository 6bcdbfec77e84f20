for ab in 0 1 2 3 4 8 12 15; do
  export EVMI_PK_ABLATE=$ab
  bash tools/prof_f32.sh abl$ab "${1:-MSD L6}"
  echo "ablate $ab: $(python tools/trace_summary.py gpurun_out/abl$ab/abl${ab}_kernel_trace.csv conv_pk_kernel | head -1)"
done
