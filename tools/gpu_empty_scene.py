import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from opengl_raytracing_amd import host, scenes, layout as L
rt = host.RayTracer(0)
sc = scenes.make_scene(2, host.generate_aabb)
# keep the lights, remove all objects
sc.objects = sc.objects[:0].copy()
rt.load(sc)
p = sc.params()
ts=[]
for k in range(60):
    rt.render(p); rt.sync(); ts.append(rt.last_kernel_ms())
print('empty scene 1080p kernel ms', np.median(ts[10:]))
sc2 = scenes.make_scene(2, host.generate_aabb)
sc2.objects = sc2.objects[-1:].copy()   # back wall only? (last object)
rt.load(sc2)
ts=[]
for k in range(60):
    rt.render(p); rt.sync(); ts.append(rt.last_kernel_ms())
print('one-plane scene kernel ms', np.median(ts[10:]), 'rays', rt.count_rays(p))
