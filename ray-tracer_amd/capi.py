"""ctypes mirror of include/rt_amd.h plus small helpers to marshal numpy scenes into it."""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    """The in-tree build; RT_AMD_LIBRARY names another build of the same library (kernel experiments)."""
    return os.environ.get("RT_AMD_LIBRARY") or os.path.join(_HERE, "csrc", "librt_amd.so")


class ImageDesc(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class MeshDesc(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p),
        ("triangles", C.c_void_p),
        ("n_vertices", C.c_uint32),
        ("n_triangles", C.c_uint32),
        ("kd", C.c_float * 3),
        ("ks", C.c_float * 3),
        ("shininess", C.c_float),
        ("transparency", C.c_float),
        ("texture", C.c_int32),
    ]


class SphereDesc(C.Structure):
    _fields_ = [
        ("center", C.c_float * 3),
        ("radius", C.c_float),
        ("kd", C.c_float * 3),
        ("ks", C.c_float * 3),
        ("shininess", C.c_float),
        ("transparency", C.c_float),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("meshes", C.POINTER(MeshDesc)),
        ("n_meshes", C.c_uint32),
        ("textures", C.POINTER(ImageDesc)),
        ("n_textures", C.c_uint32),
        ("spheres", C.POINTER(SphereDesc)),
        ("n_spheres", C.c_uint32),
        ("skybox", C.POINTER(ImageDesc)),
    ]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("v", C.c_float * 21)]


_FEATURE_NAMES = [
    "shading", "recursive", "hard_shadow", "soft_shadow", "normal_interp", "texture_mapping", "accel_structure",
    "environment_mapping", "bvh_sah_binning", "motion_blur", "bloom", "bilinear_filtering", "mipmap_filtering",
    "multiple_rays", "glossy", "transparency", "depth_of_field",
]


class Features(C.Structure):
    _fields_ = [(n, C.c_int32) for n in _FEATURE_NAMES]


def features(**kw):
    """Features with accel_structure on by default, everything else off unless named."""
    f = Features()
    f.accel_structure = 1
    for k, v in kw.items():
        if k not in _FEATURE_NAMES:
            raise KeyError(k)
        setattr(f, k, int(v))
    return f


class Params(C.Structure):
    _fields_ = [
        ("samples_per_unit", C.c_int32),
        ("samples_per_unit_parallel", C.c_int32),
        ("ray_multiplier", C.c_int32),
        ("ray_depth", C.c_int32),
        ("glossy_filter_size", C.c_int32),
        ("glossy_sigma", C.c_float),
        ("dof_samples", C.c_int32),
        ("dof_aperture", C.c_float),
        ("dof_f", C.c_float),
        ("dof", C.c_float),
        ("bloom_threshold", C.c_float),
        ("bloom_size", C.c_int32),
        ("bloom_sigma", C.c_float),
        ("bloom_debug", C.c_int32),
        ("bloom_gaussian", C.c_int32),
        ("bloom_scale", C.c_float),
        ("enabled_skybox", C.c_int32),
        ("enabled_refl_map", C.c_int32),
        ("max_transparency_depth", C.c_int32),
        ("seed", C.c_uint32),
        ("count_tests", C.c_int32),
    ]


def default_params(**kw):
    """The reference's default tunables (see include/rt_amd.h for the file:line of each)."""
    p = Params(10, 10, 1, 4, 64, 1.0, 3, 1.4, 2.5, 0.0, 0.7, 2, 1.0, 0, 1, 1.0, 0, 0, 16, 1, 0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


class Camera(C.Structure):
    _fields_ = [
        ("fovy", C.c_float),
        ("aspect", C.c_float),
        ("look_at", C.c_float * 3),
        ("distance", C.c_float),
        ("rotation", C.c_float * 3),
    ]


def _f32(x):
    return float(np.float32(x))


def make_camera(fov_deg, distance, look_at, rot_deg, width, height):
    """Camera from the reference's CameraConfig units (degrees); conversions are done in fp32 like glm::radians."""
    c = Camera()
    rad = np.float32(0.01745329251994329576923690768489)
    c.fovy = _f32(np.float32(fov_deg) * rad)
    c.aspect = _f32(np.float32(width) / np.float32(height))
    c.look_at = (C.c_float * 3)(*[_f32(v) for v in look_at])
    c.distance = _f32(distance)
    c.rotation = (C.c_float * 3)(*[_f32(np.float32(v) * rad) for v in rot_deg])
    return c


class Ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("t", C.c_float)]


class Hit(C.Structure):
    _fields_ = [
        ("hit", C.c_int32),
        ("mesh", C.c_int32),
        ("triangle", C.c_int32),
        ("t", C.c_float),
        ("normal", C.c_float * 3),
        ("bary", C.c_float * 3),
        ("kd", C.c_float * 3),
    ]


RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("t", "<f4")])
HIT_DTYPE = np.dtype([("hit", "<i4"), ("mesh", "<i4"), ("triangle", "<i4"), ("t", "<f4"), ("normal", "<f4", 3), ("bary", "<f4", 3), ("kd", "<f4", 3)])
assert RAY_DTYPE.itemsize == C.sizeof(Ray) == 28
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 52


class NodeRecord(C.Structure):
    _fields_ = [("is_leaf", C.c_int32), ("level", C.c_int32), ("lower", C.c_float * 3), ("upper", C.c_float * 3), ("a", C.c_int32), ("b", C.c_int32)]


NODE_DTYPE = np.dtype([("is_leaf", "<i4"), ("level", "<i4"), ("lower", "<f4", 3), ("upper", "<f4", 3), ("a", "<i4"), ("b", "<i4")])
PRIM_DTYPE = np.dtype([("triangle", "<i4"), ("mesh", "<i4")])
assert NODE_DTYPE.itemsize == C.sizeof(NodeRecord) == 40


class BvhOptions(C.Structure):
    _fields_ = [("sah_binning", C.c_int32), ("max_level", C.c_int32), ("sah_bins", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("bounce_rays", C.c_uint64),
        ("probe_rays", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("triangle_tests", C.c_uint64),
        ("transparency_overflows", C.c_uint64),
        ("kernel_ms", C.c_float),
        ("resolve_ms", C.c_float),
        ("shadow_box_tests", C.c_uint64),
        ("shadow_triangle_tests", C.c_uint64),
        ("primary_ms", C.c_float),
        ("shadow_ms", C.c_float),
        ("shade_ms", C.c_float),
        ("cull_ms", C.c_float),
        ("culled_shadow_rays", C.c_uint64),
        ("wavefront_nodes", C.c_uint64),
        ("stream_packets", C.c_uint64),
        ("stream_box_steps", C.c_uint64),
        ("stream_prim_steps", C.c_uint64),
        ("stream_entries", C.c_uint64),
        ("shadow_b_ms", C.c_float),
    ]

    def total_rays(self):
        """Queries actually traced."""
        return self.primary_rays + self.shadow_rays + self.bounce_rays + self.probe_rays

    def reference_rays(self):
        """Queries the reference's algorithm issues for the same frame (traced + culled)."""
        return self.total_rays() + self.culled_shadow_rays


class SceneData:
    """A scene as numpy arrays (kept alive here) and its rt_scene_desc view.

    ``meshes``: list of dicts with vertices [n,8] f32, triangles [m,3] u32, kd, ks, shininess,
    transparency, texture.  ``textures`` / ``skybox``: uint8 arrays [h,w,3].  ``lights``: [n,22] f32
    (type, 21 payload floats).
    """

    def __init__(self, meshes, textures=(), spheres=None, skybox=None, lights=None):
        self.meshes = []
        for m in meshes:
            mm = dict(m)
            mm["vertices"] = np.ascontiguousarray(m["vertices"], dtype=np.float32).reshape(-1, 8)
            mm["triangles"] = np.ascontiguousarray(m["triangles"], dtype=np.uint32).reshape(-1, 3)
            self.meshes.append(mm)
        self.textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
        self.spheres = np.zeros((0, 12), np.float32) if spheres is None else np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 12)
        self.skybox = None if skybox is None else [np.ascontiguousarray(f, dtype=np.uint8) for f in skybox]
        self.lights = np.zeros((0, 22), np.float32) if lights is None else np.ascontiguousarray(lights, dtype=np.float32).reshape(-1, 22)
        self._build_desc()

    @classmethod
    def from_golden(cls, npz, skybox=True):
        table = npz["scene.mesh_table"]
        verts = npz["scene.vertices"]
        tris = npz["scene.triangles"]
        mats = npz["scene.materials"]
        meshes = []
        for i, (vo, vc, to, tc, tex) in enumerate(table):
            meshes.append(dict(vertices=verts[vo:vo + vc], triangles=tris[to:to + tc], kd=mats[i, 0:3], ks=mats[i, 3:6],
                               shininess=mats[i, 6], transparency=mats[i, 7], texture=int(tex)))
        textures = []
        while f"scene.texture{len(textures)}" in npz:
            textures.append(npz[f"scene.texture{len(textures)}"])
        sky = [npz["scene.skybox"][f] for f in range(6)] if skybox else None
        return cls(meshes, textures, npz["scene.spheres"], sky, npz["scene.lights"])

    def _build_desc(self):
        n = len(self.meshes)
        self._mesh_arr = (MeshDesc * max(n, 1))()
        for i, m in enumerate(self.meshes):
            d = self._mesh_arr[i]
            d.vertices = m["vertices"].ctypes.data
            d.triangles = m["triangles"].ctypes.data
            d.n_vertices = m["vertices"].shape[0]
            d.n_triangles = m["triangles"].shape[0]
            d.kd = (C.c_float * 3)(*[float(v) for v in m["kd"]])
            d.ks = (C.c_float * 3)(*[float(v) for v in m["ks"]])
            d.shininess = float(m["shininess"])
            d.transparency = float(m["transparency"])
            d.texture = int(m.get("texture", -1))
        self._tex_arr = (ImageDesc * max(len(self.textures), 1))()
        for i, t in enumerate(self.textures):
            self._tex_arr[i] = ImageDesc(t.ctypes.data, t.shape[1], t.shape[0])
        ns = self.spheres.shape[0]
        self._sph_arr = (SphereDesc * max(ns, 1))()
        for i in range(ns):
            s = self.spheres[i]
            self._sph_arr[i] = SphereDesc((C.c_float * 3)(*s[0:3]), float(s[3]), (C.c_float * 3)(*s[4:7]), (C.c_float * 3)(*s[7:10]), float(s[10]), float(s[11]))
        self.desc = SceneDesc()
        self.desc.meshes = C.cast(self._mesh_arr, C.POINTER(MeshDesc))
        self.desc.n_meshes = n
        self.desc.textures = C.cast(self._tex_arr, C.POINTER(ImageDesc))
        self.desc.n_textures = len(self.textures)
        self.desc.spheres = C.cast(self._sph_arr, C.POINTER(SphereDesc))
        self.desc.n_spheres = ns
        if self.skybox is not None:
            self._sky_arr = (ImageDesc * 6)()
            for i, f in enumerate(self.skybox):
                self._sky_arr[i] = ImageDesc(f.ctypes.data, f.shape[1], f.shape[0])
            self.desc.skybox = C.cast(self._sky_arr, C.POINTER(ImageDesc))
        else:
            self.desc.skybox = None

    def write_binary(self, path):
        """Writes the flat scene file the C++ facade's loadSceneBinary reads (magic RTSCENE1)."""
        import struct

        def image(f, img):
            f.write(struct.pack("<ii", img.shape[1], img.shape[0]))
            raw = img.tobytes()
            f.write(raw + b"\0" * (-len(raw) % 4))

        with open(path, "wb") as f:
            f.write(b"RTSCENE1")
            f.write(struct.pack("<8I", len(self.meshes), len(self.textures), self.spheres.shape[0], self.lights.shape[0],
                                1 if self.skybox is not None else 0, 0, 0, 0))
            for m in self.meshes:
                f.write(struct.pack("<II", m["vertices"].shape[0], m["triangles"].shape[0]))
                f.write(struct.pack("<8f", *[float(v) for v in m["kd"]], *[float(v) for v in m["ks"]], float(m["shininess"]), float(m["transparency"])))
                f.write(struct.pack("<i", int(m.get("texture", -1))))
                f.write(m["vertices"].tobytes())
                f.write(m["triangles"].tobytes())
            for t in self.textures:
                image(f, t)
            f.write(self.spheres.tobytes())
            f.write(self.lights.tobytes())
            if self.skybox is not None:
                for face in self.skybox:
                    image(f, face)

    def light_array(self, lights=None):
        src = self.lights if lights is None else np.asarray(lights, dtype=np.float32).reshape(-1, 22)
        arr = (Light * max(len(src), 1))()
        for i, row in enumerate(src):
            arr[i].type = int(row[0])
            arr[i].v = (C.c_float * 21)(*[float(v) for v in row[1:22]])
        return arr, len(src)

    @property
    def n_triangles(self):
        return sum(m["triangles"].shape[0] for m in self.meshes)


class RtError(RuntimeError):
    pass


class Library:
    """Thin object wrapper over librt_amd.so.  Raises if the library (the HIP path) is not built."""

    def __init__(self, path=None):
        path = path or library_path()
        if not os.path.exists(path):
            raise RtError(f"{path} is missing: build it with __graft_entry__.build(); there is no CPU fallback")
        self.lib = lib = C.CDLL(path)
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_init.argtypes = [C.c_int32]
        lib.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        lib.rt_scene_destroy.argtypes = [C.c_void_p]
        lib.rt_scene_destroy.restype = None
        lib.rt_bvh_build.argtypes = [C.c_void_p, C.POINTER(BvhOptions), C.POINTER(C.c_void_p)]
        lib.rt_bvh_build_device.argtypes = [C.c_void_p, C.POINTER(BvhOptions), C.POINTER(C.c_void_p)]
        lib.rt_bvh_device_sorted.argtypes = [C.c_void_p]
        lib.rt_bvh_destroy.argtypes = [C.c_void_p]
        lib.rt_bvh_destroy.restype = None
        for fn in ("rt_bvh_num_levels", "rt_bvh_num_leaves", "rt_bvh_num_nodes", "rt_bvh_num_prim_refs"):
            getattr(lib, fn).argtypes = [C.c_void_p]
        lib.rt_bvh_export.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        lib.rt_intersect_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Features), C.c_void_p, C.c_int32, C.c_void_p]
        lib.rt_visibility_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.POINTER(Features), C.c_void_p]
        lib.rt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Light), C.c_int32, C.POINTER(Features), C.POINTER(Params),
                                  C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Stats)]
        lib.rt_render_blocks.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Light), C.c_int32, C.POINTER(Features), C.POINTER(Params),
                                         C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Stats)]
        lib.rt_bloom.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Params), C.c_int32, C.c_void_p]
        lib.rt_bloom_static.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Params), C.c_void_p]
        lib.rt_bloom_static_max_pixels.restype = C.c_uint64
        lib.rt_bloom_static_max_pixels.argtypes = []
        lib.rt_default_params.argtypes = [C.POINTER(Params)]
        lib.rt_default_params.restype = None
        lib.rt_comm_unique_id.argtypes = [C.c_void_p]
        lib.rt_comm_init_rank.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        lib.rt_comm_init_all.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.rt_comm_destroy.argtypes = [C.c_void_p]
        lib.rt_comm_destroy.restype = None
        lib.rt_gather_blocks.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        lib.rt_render_blocks_gather.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Light), C.c_int32, C.POINTER(Features), C.POINTER(Params),
                                                C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Stats)]

    def check(self, status):
        if status != 0:
            raise RtError(f"rt_amd status {status}: {self.lib.rt_last_error().decode()}")

    def init(self, device=0):
        self.check(self.lib.rt_init(device))

    def scene_create(self, scene: SceneData):
        h = C.c_void_p()
        self.check(self.lib.rt_scene_create(C.byref(scene.desc), C.byref(h)))
        return h

    def bvh_build(self, scene_handle, sah=False, max_level=32, sah_bins=64):
        h = C.c_void_p()
        opt = BvhOptions(int(sah), max_level, sah_bins)
        self.check(self.lib.rt_bvh_build(scene_handle, C.byref(opt), C.byref(h)))
        return h

    def bvh_build_device(self, scene_handle, sah=False, max_level=32, sah_bins=64):
        """rt_bvh_build_device: the same hierarchy built on the GPU (host build for the SAH variant and a few other
        cases); returns (handle, built_on_device)."""
        h = C.c_void_p()
        opt = BvhOptions(int(sah), max_level, sah_bins)
        self.check(self.lib.rt_bvh_build_device(scene_handle, C.byref(opt), C.byref(h)))
        return h, bool(self.lib.rt_bvh_device_sorted(h))

    def bvh_export(self, bvh):
        nn = self.lib.rt_bvh_num_nodes(bvh)
        npr = self.lib.rt_bvh_num_prim_refs(bvh)
        nodes = np.zeros(nn, NODE_DTYPE)
        prims = np.zeros(npr, PRIM_DTYPE)
        self.check(self.lib.rt_bvh_export(bvh, nodes.ctypes.data, nn, prims.ctypes.data, npr))
        return nodes, prims

    def intersect(self, bvh, rays, feats):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], HIT_DTYPE)
        self.check(self.lib.rt_intersect_batch(bvh, rays.ctypes.data, rays.shape[0], C.byref(feats), hits.ctypes.data, 0, None))
        return hits

    def visibility(self, bvh, rays, samples, feats):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((rays.shape[0], samples.shape[0]), np.uint8)
        self.check(self.lib.rt_visibility_batch(bvh, rays.ctypes.data, rays.shape[0], samples.ctypes.data, samples.shape[0], C.byref(feats), out.ctypes.data))
        return out

    def render(self, bvh, camera, lights, n_lights, feats, params, width, height, row_begin=0, row_end=None, out=None, device_ptr=0, stream=None):
        """Renders rows [row_begin,row_end); returns (image [rows,width,3] f32, Stats) for host output."""
        row_end = height if row_end is None else row_end
        stats = Stats()
        if out is None:
            img = np.zeros((row_end - row_begin, width, 3), np.float32)
            ptr = img.ctypes.data
        else:
            img, ptr = None, out
        self.check(self.lib.rt_render(bvh, C.byref(camera), lights, n_lights, C.byref(feats), C.byref(params), width, height,
                                      row_begin, row_end, ptr, device_ptr, stream, C.byref(stats)))
        return img, stats

    def render_blocks(self, bvh, camera, lights, n_lights, feats, params, width, height, row_block, first_block, block_stride,
                      out=None, device_ptr=0, stream=None):
        """Renders Screen-row blocks first_block, first_block + block_stride, ... of row_block rows each in one call;
        returns (image [rows of those blocks, width, 3] f32, Stats) for host output."""
        n_blocks = (height + row_block - 1) // row_block
        rows = sum(min(height, (b + 1) * row_block) - b * row_block for b in range(first_block, n_blocks, block_stride))
        stats = Stats()
        if out is None:
            img = np.zeros((rows, width, 3), np.float32)
            ptr = img.ctypes.data
        else:
            img, ptr = None, out
        self.check(self.lib.rt_render_blocks(bvh, C.byref(camera), lights, n_lights, C.byref(feats), C.byref(params), width, height,
                                             row_block, first_block, block_stride, ptr, device_ptr, stream, C.byref(stats)))
        return img, stats

    def bloom(self, image, params):
        img = np.ascontiguousarray(image, dtype=np.float32)
        h, w = img.shape[0], img.shape[1]
        self.check(self.lib.rt_bloom(img.ctypes.data, w, h, C.byref(params), 0, None))
        return img

    # ---- frame assembly across GPUs (RCCL under the C ABI)
    def comm_unique_id(self):
        """rt_comm_unique_id: 128 bytes that rank 0 hands to the other ranks by whatever channel there is."""
        buf = (C.c_uint8 * 128)()
        self.check(self.lib.rt_comm_unique_id(buf))
        return bytes(buf)

    def comm_init_rank(self, unique_id, rank, world):
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self.check(self.lib.rt_comm_init_rank(buf, rank, world, C.byref(h)))
        return h

    def comm_destroy(self, comm):
        self.lib.rt_comm_destroy(comm)

    def gather_blocks(self, comm, root, share_ptr, frame_ptr, width, height, row_block, stream=None):
        """rt_gather_blocks: every rank's row blocks (device memory) into the root's device frame; queued on `stream`."""
        self.check(self.lib.rt_gather_blocks(comm, root, share_ptr, frame_ptr, width, height, row_block, stream))

    def bloom_device(self, ptr, width, height, params, stream=None):
        """rt_bloom in place on a device frame (queued on `stream`, no wait)."""
        self.check(self.lib.rt_bloom(ptr, width, height, C.byref(params), 1, stream))

    def bloom_static(self, ptr, width, height, params, stream=None):
        """rt_bloom_static: the same on a device frame without any allocation (frame size is limited)."""
        self.check(self.lib.rt_bloom_static(ptr, width, height, C.byref(params), stream))
