#!/bin/bash
# The wavefront-per-residue message pass against its 8-wavefront form, and what its parts cost, in ONE gpurun call (round 5,
# docs/NOTEBOOK.md 9.10-9.11). All variants are debug builds of tmpnn_msg.hip only; results of the ablations are wrong by construction.
#   tools/ab_msg_wave.sh build      (here, no GPU needed)   -> thermompnn_amd/libtmpnn_mw_*.so
#   tools/ab_msg_wave.sh run        (on the GPU box)        -> gpurun_out/${TAG:-r06}_ab_msg_wave.txt
# Per variant: per-kernel HIP-event times of the bench batch (tools/ab_time.py) and, from the kernel's own counters, the shader clock
# it ran at (TMPNN_MSG_PROF=1: cycle counter against the 100 MHz reference around one wavefront's loop, after 40 forwards).
# and the loop cycles of all eight wavefronts of the first and the last workgroup (round 6: the 11 : 5 residue split, NOTEBOOK 10.3f).
cd "$(dirname "$0")/.."
L=thermompnn_amd/libtmpnn_mw
case "$1" in
build)
  for v in "shipped" "nolds -DTM_ABL_WAVE_NOLDS=1" "nogelu -DTM_ABL_NOGELU=1" "nomfma -DTM_ABL_NOMFMA=1" "nogelu_nomfma -DTM_ABL_NOGELU=1 -DTM_ABL_NOMFMA=1"; do
    set -- $v; n=$1; shift
    python -m thermompnn_amd.build --variant mw_$n -DTMPNN_DEBUG_BUILD "$@" --only=tmpnn_msg.hip 2>&1 | tail -1 &
  done
  wait ;;
run)
  mkdir -p gpurun_out
  { echo "# variant: ms/step, per-kernel ms (tools/ab_time.py); then the wave kernel's phase timers + clock"
    echo "## 8-wavefront form (TMPNN_MSG_WAVE_MIN=1000000: the launcher never takes the wavefront-per-residue path; same bits)"
    TMPNN_LIB=${L}_shipped.so TMPNN_MSG_WAVE_MIN=1000000 python tools/ab_time.py 8wavefront
    for n in shipped nolds nogelu nomfma nogelu_nomfma; do
      echo "## $n"
      TMPNN_LIB=${L}_$n.so python tools/ab_time.py $n
      TMPNN_LIB=${L}_$n.so TMPNN_MSG_PROF=1 python tools/prof_run.py 2>&1 | grep -E "wave (phases|loops)" | tail -2
    done
    echo "## 8-wavefront form again (alternation)"
    TMPNN_LIB=${L}_shipped.so TMPNN_MSG_WAVE_MIN=1000000 python tools/ab_time.py 8wavefront
    echo "## shipped again"
    TMPNN_LIB=${L}_shipped.so python tools/ab_time.py shipped; } 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/${TAG:-r06}_ab_msg_wave.txt ;;
*) echo "usage: $0 build|run" ;;
esac
