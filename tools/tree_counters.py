"""Work counters per ray of config 3 on the uploaded TLAS and on the device-built one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilgpu_raytracing_amd import _types as T, scenes, engine
r = engine.RTRenderer([0]); cfg = scenes.CONFIGS[3]
s = engine.Scene(); scenes.build(3, s); r.commit(s)
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=4)


def show(tag):
    st = r.render_params(p, None, flags=T.FLAG_COUNTERS)
    for i in range(2):
        d = st.k[i].as_dict(); rays = d["rays_closest"] + d["rays_shadow"]
        print(tag, "launch", i, "rays", rays, "node_visits/ray %.1f" % (d["node_visits"] / max(1, rays)),
              "leaf_instances/ray %.2f" % (d["leaf_instances"] / max(1, rays)), "sphere_tests/ray %.2f" % (d["sphere_tests"] / max(1, rays)))


show("uploaded tree    ")
r.update_instances([], [], T.REBUILD_FORCE_REBUILD)
show("device-built tree")
