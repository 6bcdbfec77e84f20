for ab in ${ABL:-0 15 31}; do
  export EVMI_PK_ABLATE=$ab
  bash tools/prof_f32.sh abl$ab "${1:-MSD L6}"
  echo "ablate $ab: $(python tools/trace_summary.py gpurun_out/abl$ab/abl${ab}_kernel_trace.csv conv_pk_kernel | head -1)"
done
