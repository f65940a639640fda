"""Scenes and the pose-sweep checker for the GJK / EPA narrowphase tests (hostsim on CPU, HIP on the GPU).

A sweep puts one free body in N random / axis-aligned / nearly aligned poses around a fixed geom,
runs mj_forward on both sides and compares the contact lists: count, geom ids, distance, position
and frame.  mjc_Convex (engine_collision_convex.c:881) is a chain of data-dependent branches
(simplex cases, EPA horizon walks, face-clipping cases); the sweep is what exercises them all.
"""
import numpy as np

CUBELET = """8.075 9.5 -8.075 -8.075 9.5 -8.075 8.075 9.5 8.075 -8.075 9.5 8.075 -9.5 8.075 -8.075 -9.5 -8.075 -8.075
 -9.5 8.075 8.075 -9.5 -8.075 8.075 8.075 -9.5 -8.075 8.075 -9.5 8.075 -8.075 -9.5 -8.075 -8.075 -9.5 8.075
 9.5 8.075 8.075 9.5 -8.075 8.075 9.5 8.075 -8.075 9.5 -8.075 -8.075 8.075 8.075 9.5 -8.075 8.075 9.5
 8.075 -8.075 9.5 -8.075 -8.075 9.5 8.075 -8.075 -9.5 -8.075 -8.075 -9.5 8.075 8.075 -9.5 -8.075 8.075 -9.5"""
_PHI = (1 + 5**.5)/2
ICO = " ".join(f"{x} {y} {z}" for x, y, z in [(-1, _PHI, 0), (1, _PHI, 0), (-1, -_PHI, 0), (1, -_PHI, 0), (0, -1, _PHI), (0, 1, _PHI),
                                              (0, -1, -_PHI), (0, 1, -_PHI), (_PHI, 0, -1), (_PHI, 0, 1), (-_PHI, 0, -1), (-_PHI, 0, 1)])
TET = "1 1 1  1 -1 -1  -1 1 -1  -1 -1 1"
ASSET = f"""<asset>
 <mesh name="cubelet" scale="1e-2 1e-2 1e-2" vertex="{CUBELET}"/>
 <mesh name="ico" scale=".08 .08 .08" vertex="{ICO}"/>
 <mesh name="tet" scale=".1 .1 .1" vertex="{TET}"/>
</asset>"""

GEOMS = {
    "box": 'type="box" size=".2 .15 .1"', "cyl": 'type="cylinder" size=".12 .15"', "cap": 'type="capsule" size=".08 .15"',
    "ell": 'type="ellipsoid" size=".2 .12 .08"', "sph": 'type="sphere" size=".12"',
    "cub": 'type="mesh" mesh="cubelet"',      # 24 vertices: hill-climbing support (mjMESH_HILLCLIMB_MIN = 10)
    "ico": 'type="mesh" mesh="ico"',          # 12 vertices, triangles only
    "tet": 'type="mesh" mesh="tet"',          # 4 vertices: exhaustive support
    "pla": 'type="plane" size="1 1 .1"',
}

# the cells of mjCOLLISIONFUNC (engine_collision_driver.c:45-56) that go to mjc_Convex / mjc_PlaneConvex
PRIMITIVE_PAIRS = [("box", "cyl"), ("cyl", "cyl"), ("cyl", "cap"), ("ell", "ell"), ("ell", "box"), ("ell", "cyl"), ("ell", "cap"),
                   ("ell", "sph"), ("pla", "ell")]
MESH_PAIRS = [("cub", "cub"), ("cub", "box"), ("ico", "cub"), ("tet", "cub"), ("tet", "tet"), ("ico", "ico"), ("cub", "cap"),
              ("cub", "cyl"), ("sph", "cub"), ("ell", "ico"), ("pla", "cub"), ("pla", "tet")]


def scene(fixed, moving, margin=0.0, aligned=False):
    euler = "" if aligned else 'euler="10 20 15"'
    return f"""
<mujoco><default><geom margin="{margin}"/></default>{ASSET}<worldbody>
  <geom name="fixed" {GEOMS[fixed]} pos="0 0 .3" {euler}/>
  <body pos="0 0 .5"><freejoint/><geom {GEOMS[moving]} condim="3"/></body>
</worldbody></mujoco>"""


def poses(N, seed, aligned=False):
    rng = np.random.default_rng(seed)
    qpos = np.zeros((N, 7))
    for k in range(N):
        q = rng.normal(size=4)
        if k % 5 == 1: q = np.array([1, 0, 0, 0.]) + rng.normal(size=4)*1e-3
        if k % 5 == 2: q = np.array([1, 0, 0, 0.])
        if k % 5 == 3: q = np.array([1, 0, 0, 1.])
        qpos[k, :3] = [rng.uniform(-.3, .3), rng.uniform(-.3, .3), rng.uniform(.05, .6)]
        if aligned and k % 2: qpos[k, :2] = np.round(qpos[k, :2]*10)/10
        qpos[k, 3:] = q/np.linalg.norm(q)
    return qpos


def sweep(rb, K, lib, xml_path, N, seed, aligned=False, tol=0.0):
    """returns (histogram of reference contact counts, number of poses whose contact list differs)"""
    m = rb.MjModel.from_xml_path(str(xml_path))
    dm = K.DeviceModel(lib, m, 16, 64)
    d = rb.MjData(m)
    qpos = poses(N, seed, aligned)
    b = K.Batch(dm, N)
    b.reset()
    b.set("qpos", qpos)
    b.forward()
    assert b.get("warning").sum() == 0
    counts = b.get("counts")[:, 0]
    cd = b.get("con_dist"); cp = b.get("con_pos").reshape(N, -1, 3); cf = b.get("con_frame").reshape(N, -1, 9)
    cg = b.get("con_geom").reshape(N, -1, 2)
    hist, bad = {}, 0
    for k in range(N):
        rb.mj_resetData(m, d)
        d.qpos[:] = qpos[k]
        rb.mj_forward(m, d)
        n = d.ncon
        hist[n] = hist.get(n, 0) + 1
        rc = d.contact[:n]
        ok = counts[k] == n and np.array_equal(cg[k, :n], rc["geom"])
        if ok and n:
            if tol == 0:
                ok = np.array_equal(cd[k, :n], rc["dist"]) and np.array_equal(cp[k, :n], rc["pos"]) and np.array_equal(cf[k, :n], rc["frame"])
            else:
                ok = (np.abs(cd[k, :n] - rc["dist"]).max() <= tol and np.abs(cp[k, :n] - rc["pos"]).max() <= tol and
                      np.abs(cf[k, :n] - rc["frame"]).max() <= tol)
        bad += 0 if ok else 1
    return hist, bad
