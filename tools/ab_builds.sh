#!/bin/bash
# A/B of several builds of the library on ONE box (one gpurun call):  bash tools/ab_builds.sh OUTDIR "name=libpath ..." "WL ..."
# e.g. make -C metdetpy_amd/csrc -j8 OUT=../../_ab/libu2.so OBJ=/tmp/ab_u2 EXTRA=-DWIDE_U16_BIGN=2;  bash tools/ab_builds.sh r04_21 "def=metdetpy_amd/libmetdet_hip.so u2=metdetpy_amd/libmetdet_hip_u2.so" "C3"
# Every build is loaded through MDH_LIB; two alternating repetitions per workload; prints frames/s, the isolated launch, frac, frac at pace, stand-alone spans.
O=gpurun_out/$1; mkdir -p $O
C="--no-pmc --cpu-seconds 0 --parity-frames 0 --standalone-passes 12 --no-secondary --steps 5 --warmup 1"
for rep in 1 2; do
 for wl in $3; do
  for v in $2; do
   name=${v%%=*}; lib=${v#*=}
   MDH_LIB=$PWD/$lib python bench.py --workload $wl $C > $O/${wl}_${name}_$rep.json 2>>$O/err
  done
 done
done
python - $O <<'P'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'ERR'); continue
    r=d['roofline']; print(f.split('/')[-1], int(d['value']), round(r.get('launch_ms',0),3), round(r['frac'],3), round(r['frac_at_pipeline_pace'],3), {k:round(v,3) for k,v in d.get('kernels_ms_standalone',{}).items()})
P
