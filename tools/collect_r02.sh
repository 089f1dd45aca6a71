#!/bin/bash
# Runs ON THE GPU BOX (through gpurun). Round-2 evidence the round-1 review asked for:
#   probe      which independent H.264 decoders exist on the box (none expected)
#   w1         one 1080p stream, frames back to back (CEDAR_ROW_PIPE=0): kernel durations = work
#   s8 / s32   8 and 32 lock-step streams
# Each leg: rocprofv3 --kernel-trace --stats of bench.py, the stats CSV and a text summary of
# the trace (tools/trace_summary.py). Outputs under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
{
  echo "== binaries"; for b in ffmpeg ffprobe ffplay x264 gst-launch-1.0 mplayer vlc mpv; do printf "%s: " $b; command -v $b || echo none; done
  echo "== shared objects"; ldconfig -p | grep -E 'avcodec|avformat|openh264|x264|rocdecode|libva|vdpau|gstreamer|libde265|libdav1d' || echo none
  echo "== /opt/rocm"; ls /opt/rocm/lib | grep -i -E 'decode|jpeg|vcn|va' || echo none
  echo "== python modules"
  for m in av cv2 torchvision torchcodec imageio imageio_ffmpeg decord ffmpeg skvideo; do python3 -c "import $m; print('$m', getattr($m,'__version__','?'))" 2>/dev/null || echo "$m: not importable"; done
  echo "== dri"; ls -l /dev/dri 2>&1
} > $OUT/decoder_probe.txt 2>&1
cd /tmp
leg() { # name, env assignment (may be empty), bench args...
  local name=$1 envs=$2; shift 2
  ( [ -n "$envs" ] && export $envs
    timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- \
      python3 $R/bench.py --cpu-frames 0 --batched-streams 0 "$@" > $OUT/$name.json 2> $OUT/$name.err ) || return 1
  local tr=$(find $OUT/$name -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_summary.py $tr > $OUT/${name}_trace_summary.txt 2>&1
  cp $(find $OUT/$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
  echo "leg $name done"
}
leg w1 CEDAR_ROW_PIPE=0 --steps 50 --warmup 25 &&
leg p1 "" --steps 100 --warmup 25 &&
leg s8 "" --streams 8 --steps 50 --warmup 25 &&
leg s32 "" --streams 32 --steps 50 --warmup 25 &&
echo collected $TAG
