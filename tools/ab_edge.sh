#!/bin/bash
# The edge update's ablation ladder in ONE gpurun call (round 6): what the 10.5 k-cycle tile of the 8-wavefront form is made of, in STEP
# time, kernel time, clock and cycles per tile — and the same ladder for the wavefront-per-block form (tmpnn_edge_wave.hip) beside it.
# Every variant is a debug build of tmpnn_edge.hip + tmpnn_edge_wave.hip (+ tmpnn_api.hip: the K-permuted images of the edge weights) only; results of the ablations are wrong by construction.
#   tools/ab_edge.sh build      (here, no GPU needed)   -> thermompnn_amd/libtmpnn_ew_*.so
#   tools/ab_edge.sh run        (on the GPU box)        -> gpurun_out/${TAG:-r06}_ab_edge.txt
# Per variant and form: ms/step + per-kernel HIP-event times of the bench batch (tools/ab_time.py) and, from the kernel's own counters
# (TMPNN_EDGE_PROF=1, after 40 forwards), its phases in cycles, the shader clock it ran at and the cycles per tile.
# TMPNN_EDGE_WAVE_MIN=-1 (debug library only) keeps the launcher on the 8-wavefront form.
cd "$(dirname "$0")/.."
L=thermompnn_amd/libtmpnn_ew
VARS="shipped nomfma nogelu noln nosplit none"
case "$1" in
build)
  for v in "shipped" "nomfma -DTM_ABL_NOMFMA=1" "nogelu -DTM_ABL_NOGELU=1" "noln -DTM_ABL_NOLN=1" "nosplit -DTM_ABL_NOSPLIT=1" \
           "none -DTM_ABL_NOMFMA=1 -DTM_ABL_NOGELU=1 -DTM_ABL_NOLN=1 -DTM_ABL_NOSPLIT=1"; do
    set -- $v; n=$1; shift
    python -m thermompnn_amd.build --variant ew_$n -DTMPNN_DEBUG_BUILD "$@" --only=tmpnn_edge.hip --only=tmpnn_edge_wave.hip --only=tmpnn_api.hip 2>&1 | tail -1 &
  done
  wait ;;
run)
  mkdir -p gpurun_out
  one() {   # $1 = variant, $2 = form (8wf | wave)
    local env="TMPNN_EDGE_WAVE_MIN=6"; [ "$2" = 8wf ] && env="TMPNN_EDGE_WAVE_MIN=-1"
    env TMPNN_LIB=${L}_$1.so $env python tools/ab_time.py "$1/$2"
    env TMPNN_LIB=${L}_$1.so $env TMPNN_EDGE_PROF=1 python tools/prof_run.py 2>&1 | grep "enc_edge.*phases" | tail -1
  }
  { echo "# variant/form: ms/step, per-kernel ms (tools/ab_time.py); then the edge kernel's phase timers, clock, cycles per tile"
    for rep in 1 2; do
      echo "## alternation $rep"
      for n in $VARS; do one $n 8wf; one $n wave; done
    done; } 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/${TAG:-r06}_ab_edge.txt ;;
*) echo "usage: $0 build|run" ;;
esac
