"""Print the key figures of bench.py JSON lines:  python tools/sumbench.py gpurun_out/r02d/*.json"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", open(f.replace(".json", ".err")).read()[-300:].replace("\n", " | ")); continue
    r = d.get('roofline', {})
    print(f"{f.split('/')[-1]:26s} {d['value']:10.0f} fps {d['ms_per_step']:.3f} ms/step B {d['config']['frames_per_step']} frac(alone) {r.get('frac', 0):.3f} in-pipe/launch {r.get('frac_in_pipeline_per_launch', 0):.3f} pace {r.get('frac_at_pipeline_pace', 0):.3f} {d.get('parity', {}).get('verdict')}")
    print('    pipe ', {k[:10]: round(v, 3) for k, v in d.get('kernels_ms_per_step', {}).items()})
    print('    alone', {k[:10]: round(v, 3) for k, v in d.get('kernels_ms_standalone', {}).items()})
