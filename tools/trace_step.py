import sys, os, time
sys.path.insert(0, os.getcwd())
from focalsv_amd import _lib, pipeline, synth
regions=[synth.make_region(i, start=i*60000) for i in range(256)]
inputs=[pipeline.region_from_synth(r) for r in regions]
with _lib.Context(0) as ctx:
    b=pipeline.upload_regions(ctx, inputs)
    pipeline.run_hot_path(ctx,b)
    os.environ["FSV_TRACE"]="1"
    t=time.time(); r=pipeline.run_hot_path(ctx,b); print("step", time.time()-t, r.host_ms, file=sys.stderr)
    b.free(ctx)
