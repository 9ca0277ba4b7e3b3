"""ctypes binding of include/tetsim.h (libtetsim_hip.so).  1:1 with the C ABI; no logic here.

The library is REQUIRED: importing this module without a built libtetsim_hip.so raises, and creating a
body without a HIP device raises TetSimError(ENODEVICE).  There is no CPU fallback anywhere in the product.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TETSIM_HIP_LIB") or os.path.join(_HERE, "libtetsim_hip.so")   # same override as the N-API shim

OK, EINVAL, ENODEVICE, EHIP, ENOMEM, ECOMM, ESTATE = range(7)
SOLVER_POLAR_JACOBI, SOLVER_NEOHOOKEAN_GS = 0, 1
PRECISE, FAST = 0, 1
ORDER_ORIGINAL, ORDER_COLOURED, ORDER_CLUSTERED = 0, 1, 2
FLAG_REF_SLOT_TABLE, FLAG_REF_FIXED_BOUNDS, FLAG_GATHER_FORMULATION, FLAG_CONSTANT_REST_SHAPE, FLAG_REF_GRAB_TEXEL, FLAG_DEEP_GHOSTS = 1, 2, 4, 8, 16, 32
FLAG_REF_ROTATION_EXIT = 64
FLAG_LEAN_STATE = 128
K_TET, K_VERTEX, K_HALO, K_COUNT = 0, 1, 2, 3


class TetSimParams(C.Structure):
    _fields_ = [("gravity", C.c_double), ("friction", C.c_double), ("devCompliance", C.c_double),
                ("volCompliance", C.c_double), ("worldBounds", C.c_double * 6)]


class TetSimOptions(C.Structure):
    _fields_ = [("solver", C.c_int32), ("precision", C.c_int32), ("order", C.c_int32), ("flags", C.c_uint32),
                ("device", C.c_int32), ("density", C.c_double), ("part_count", C.c_int32),
                ("part_index", C.c_int32), ("vert_owner", C.POINTER(C.c_int32)), ("tet_colour", C.POINTER(C.c_int32))]


class TetSimMeshArrays(C.Structure):
    _fields_ = [("num_particles", C.c_uint32), ("num_elems", C.c_uint32), ("num_edges", C.c_uint32),
                ("num_vis_verts", C.c_uint32), ("num_vis_tris", C.c_uint32), ("part_count", C.c_uint32),
                ("verts", C.POINTER(C.c_float)), ("tets", C.POINTER(C.c_int32)), ("edge_ids", C.POINTER(C.c_int32)),
                ("vis_verts", C.POINTER(C.c_float)), ("vis_tri_ids", C.POINTER(C.c_int32)),
                ("tet_colour", C.POINTER(C.c_int32)), ("vert_owner", C.POINTER(C.c_int32))]


class TetSimInfo(C.Structure):
    _fields_ = [("num_particles", C.c_uint32), ("num_elems", C.c_uint32), ("owned_particles", C.c_uint32),
                ("local_particles", C.c_uint32), ("local_elems", C.c_uint32), ("owned_elems", C.c_uint32),
                ("num_levels", C.c_uint32), ("max_valence", C.c_uint32), ("dropped_slots", C.c_uint32),
                ("num_neighbours", C.c_uint32), ("device_bytes", C.c_uint64), ("solver", C.c_int32),
                ("precision", C.c_int32), ("order", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("num_vis_verts", C.c_uint32), ("num_bodies", C.c_uint32), ("fused_particle_pass", C.c_uint32), ("total_vis_verts", C.c_uint32)]


class TetSimProfile(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("kernel_ms", C.c_double * K_COUNT), ("launches", C.c_uint32 * K_COUNT),
                ("substeps", C.c_uint32), ("tets_per_tet_launch", C.c_uint32)]


class TetSimLibraryInfo(C.Structure):
    _fields_ = [("abi", C.c_int32), ("ablation", C.c_int32), ("debug_env", C.c_uint32), ("source_sha", C.c_char * 20),
                ("kernel_sha", C.c_char * 20)]


DEBUG_ENV_NAMES = ["TETSIM_DEBUG_LOOPBACK_HALO", "TETSIM_DEBUG_LOOPBACK_COPY", "TETSIM_DEBUG_ONE_STREAM", "TETSIM_DEBUG_GROUP_SYNC",
                   "TETSIM_DEBUG_HOSTPROF", "TETSIM_DEBUG_TRACE", "TETSIM_HALO_SYNC", "TETSIM_HALO_GRAPH", "TETSIM_DEBUG_LOOPBACK_DELAY_US", "TETSIM_NH_QUADS", "TETSIM_FUSED_PARTICLE_PASS",
                   "TETSIM_FRAME_KERNEL", "TETSIM_FRAME_LOCAL", "TETSIM_NH_FOLD", "TETSIM_HALO_ALIGNED_TILES", "TETSIM_HALO_FOLD_WAIT", "TETSIM_QUAD", "TETSIM_QUAD_POLL_DELAY", "TETSIM_NH_FRAME",
                   "TETSIM_NH_ONE_LAUNCH", "TETSIM_PJ_ONE_LAUNCH"]


# A/B switches of settled choices: read (and reported) by the development build only (csrc/body.h: lab_env, build_info.cpp)
LAB_ENV_NAMES = ["TETSIM_DEBUG_TRACE", "TETSIM_NH_QUADS", "TETSIM_FRAME_LOCAL", "TETSIM_NH_FOLD", "TETSIM_HALO_ALIGNED_TILES", "TETSIM_QUAD_POLL_DELAY", "TETSIM_NH_FRAME"]


class TetSimCommInfo(C.Structure):
    _fields_ = [("rccl_ranks", C.c_int32), ("rccl_rank", C.c_int32), ("neighbours", C.c_uint32), ("send_bytes_per_substep", C.c_uint64),
                ("recv_bytes_per_substep", C.c_uint64), ("max_message_bytes", C.c_uint64), ("loopback", C.c_int32), ("p2p", C.c_int32)]


class TetSimPlanSizes(C.Structure):
    _fields_ = [("owned_particles", C.c_uint32), ("boundary_particles", C.c_uint32), ("local_particles", C.c_uint32),
                ("local_elems", C.c_uint32), ("owned_elems", C.c_uint32), ("num_neighbours", C.c_uint32)]


class TetSimPartQuality(C.Structure):
    _fields_ = [("owned_particles", C.c_uint32), ("ghost_particles", C.c_uint32), ("boundary_particles", C.c_uint32),
                ("local_elems", C.c_uint32), ("owned_elems", C.c_uint32), ("num_neighbours", C.c_uint32)]


class TetSimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tetsim error %d: %s" % (code, msg))
        self.code = code


# every symbol include/tetsim.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "tetsim_abi_version", "tetsim_default_options", "tetsim_default_params", "tetsim_create", "tetsim_create_batch", "tetsim_get_batch_layout", "tetsim_destroy",
    "tetsim_last_error", "tetsim_get_info", "tetsim_step", "tetsim_step_n", "tetsim_sync",
    "tetsim_library_info", "tetsim_read_quats_pinned", "tetsim_state_size", "tetsim_save_state", "tetsim_load_state",
    "tetsim_read_positions", "tetsim_read_positions_pinned", "tetsim_read_prev_positions", "tetsim_read_velocities", "tetsim_read_quats",
    "tetsim_read_vol_error", "tetsim_write_state", "tetsim_get_owned_ids", "tetsim_get_local_tets",
    "tetsim_get_tet_order", "tetsim_get_level_offsets", "tetsim_read_inv_mass", "tetsim_set_visual_mesh", "tetsim_get_visual_ids", "tetsim_halo_refresh_final", "tetsim_group_refresh_final", "tetsim_halo_probe", "tetsim_halo_p2p_probe",
    "tetsim_read_visual_mesh", "tetsim_set_visual_triangles", "tetsim_read_visual_vertex_normals", "tetsim_visual_vertex_normals_from", "tetsim_group_read_visual_vertex_normals", "tetsim_set_grab",
    "tetsim_start_grab", "tetsim_nearest_particle", "tetsim_profile", "tetsim_time_kernels", "tetsim_time_step_n", "tetsim_measure_copy_bandwidth", "tetsim_measure_stream_bandwidth",
    "tetsim_comm_unique_id", "tetsim_comm_init", "tetsim_comm_info", "tetsim_comm_selftest", "tetsim_comm_probe", "tetsim_group_step_n", "tetsim_halo_exchange_local", "tetsim_get_halo_plan",
    "tetsim_halo_p2p_export", "tetsim_halo_p2p_connect",
    "tetsim_halo_export", "tetsim_halo_import", "tetsim_prep_levels", "tetsim_prep_colours", "tetsim_prep_clusters",
    "tetsim_prep_tiles", "tetsim_prep_slot_table", "tetsim_prep_ref_grab_texels", "tetsim_prep_rest", "tetsim_prep_partition", "tetsim_prep_partition_quality", "tetsim_plan_create", "tetsim_plan_destroy", "tetsim_plan_sizes",
    "tetsim_plan_arrays", "tetsim_plan_neighbour", "tetsim_plan_neighbour_ids",
    "tetsim_plan_create_deep", "tetsim_plan_layers", "tetsim_plan_neighbour_layer2", "tetsim_plan_neighbour_layer2_ids",
    "tetsim_mesh_write", "tetsim_mesh_open", "tetsim_mesh_arrays", "tetsim_mesh_close", "tetsim_create_from_file",
]

_lib = None


def lib():
    """Load libtetsim_hip.so (building it is __graft_entry__.build()'s / tetsim_amd.build's job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libtetsim_hip.so is not built: run `python -m tetsim_amd.build` "
                          "(the HIP extension is mandatory; there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    H, fp, ip, dp = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    u32, i32, dbl = C.c_uint32, C.c_int32, C.c_double
    PP = C.POINTER(TetSimParams)
    L.tetsim_abi_version.restype = C.c_int
    L.tetsim_default_options.argtypes = [C.POINTER(TetSimOptions)]
    L.tetsim_default_options.restype = None
    L.tetsim_default_params.argtypes = [PP]
    L.tetsim_default_params.restype = None
    L.tetsim_create.argtypes = [fp, u32, ip, u32, C.POINTER(TetSimOptions), C.POINTER(H)]
    L.tetsim_create_batch.argtypes = [C.POINTER(fp), C.POINTER(u32), C.POINTER(ip), C.POINTER(u32), u32, C.POINTER(TetSimOptions), C.POINTER(H)]
    L.tetsim_get_batch_layout.argtypes = [H, C.POINTER(u32), C.POINTER(u32)]
    L.tetsim_destroy.argtypes = [H]
    L.tetsim_destroy.restype = None
    L.tetsim_comm_probe.argtypes = [H, C.c_uint64, u32, i32, u32, dp, dp]
    L.tetsim_nearest_particle.argtypes = [H, fp, ip, dp]
    L.tetsim_mesh_write.argtypes = [C.c_char_p, C.POINTER(TetSimMeshArrays)]
    L.tetsim_mesh_open.argtypes = [C.c_char_p, C.POINTER(H)]
    L.tetsim_mesh_arrays.argtypes = [H, C.POINTER(TetSimMeshArrays)]
    L.tetsim_mesh_close.argtypes = [H]
    L.tetsim_create_from_file.argtypes = [C.c_char_p, C.POINTER(TetSimOptions), C.POINTER(H)]
    L.tetsim_last_error.argtypes = [H]
    L.tetsim_last_error.restype = C.c_char_p
    L.tetsim_get_info.argtypes = [H, C.POINTER(TetSimInfo)]
    L.tetsim_step.argtypes = [H, dbl, PP]
    L.tetsim_step_n.argtypes = [H, u32, dbl, PP]
    L.tetsim_sync.argtypes = [H]
    for n in ("positions", "prev_positions", "velocities", "quats", "inv_mass"):
        getattr(L, "tetsim_read_" + n).argtypes = [H, fp]
    L.tetsim_read_positions_pinned.argtypes = [H, C.POINTER(fp)]
    L.tetsim_read_quats_pinned.argtypes = [H, C.POINTER(fp)]
    L.tetsim_library_info.argtypes = [C.POINTER(TetSimLibraryInfo)]
    L.tetsim_state_size.argtypes = [H, C.POINTER(C.c_uint64)]
    L.tetsim_save_state.argtypes = [H, C.c_void_p, C.c_uint64]
    L.tetsim_load_state.argtypes = [H, C.c_void_p, C.c_uint64]
    L.tetsim_read_vol_error.argtypes = [H, dp]
    L.tetsim_write_state.argtypes = [H, fp, fp]
    for n in ("owned_ids", "local_tets", "tet_order", "level_offsets"):
        getattr(L, "tetsim_get_" + n).argtypes = [H, ip]
    L.tetsim_set_visual_mesh.argtypes = [H, fp, u32, fp]
    L.tetsim_read_visual_mesh.argtypes = [H, fp, fp]
    L.tetsim_get_visual_ids.argtypes = [H, C.POINTER(C.c_int32)]
    L.tetsim_halo_refresh_final.argtypes = [H]
    L.tetsim_halo_probe.argtypes = [H, u32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.tetsim_halo_p2p_probe.argtypes = [H, u32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.tetsim_group_refresh_final.argtypes = [C.POINTER(H), u32]
    L.tetsim_set_visual_triangles.argtypes = [H, ip, u32]
    L.tetsim_read_visual_vertex_normals.argtypes = [H, fp]
    L.tetsim_visual_vertex_normals_from.argtypes = [H, fp, fp]
    L.tetsim_group_read_visual_vertex_normals.argtypes = [C.POINTER(H), u32, fp, fp]
    L.tetsim_set_grab.argtypes = [H, i32, fp]
    L.tetsim_start_grab.argtypes = [H, fp, ip]
    L.tetsim_profile.argtypes = [H, u32, dbl, PP, C.POINTER(TetSimProfile)]
    L.tetsim_time_kernels.argtypes = [H, u32, dbl, PP, C.POINTER(TetSimProfile)]
    L.tetsim_time_step_n.argtypes = [H, u32, dbl, PP, dp]
    L.tetsim_measure_copy_bandwidth.argtypes = [i32, C.c_uint64, u32, dp]
    L.tetsim_measure_stream_bandwidth.argtypes = [i32, C.c_uint64, u32, i32, dp]
    L.tetsim_comm_unique_id.argtypes = [C.c_void_p]
    L.tetsim_comm_init.argtypes = [H, C.c_void_p, i32, i32]
    L.tetsim_comm_selftest.argtypes = [H]
    L.tetsim_comm_info.argtypes = [H, C.POINTER(TetSimCommInfo)]
    L.tetsim_group_step_n.argtypes = [C.POINTER(H), u32, u32, dbl, PP]
    L.tetsim_halo_exchange_local.argtypes = [C.POINTER(H), u32]
    L.tetsim_halo_p2p_export.argtypes = [H, C.c_void_p]
    L.tetsim_halo_p2p_connect.argtypes = [H, C.c_void_p, u32]
    L.tetsim_get_halo_plan.argtypes = [H, ip, ip, ip, ip, ip]
    L.tetsim_halo_export.argtypes = [H, u32, fp]
    L.tetsim_halo_import.argtypes = [H, u32, fp]
    L.tetsim_prep_levels.argtypes = [ip, u32, u32, ip, C.POINTER(u32)]
    L.tetsim_prep_colours.argtypes = [ip, u32, u32, ip, C.POINTER(u32)]
    L.tetsim_prep_clusters.argtypes = [ip, u32, u32, ip, ip, ip, ip, C.POINTER(u32), C.POINTER(u32)]
    L.tetsim_prep_tiles.argtypes = [fp, u32, ip, u32, C.POINTER(u32), C.POINTER(u32), u32, ip, C.POINTER(u32), C.POINTER(C.c_uint8), C.POINTER(u32)]
    L.tetsim_prep_slot_table.argtypes = [ip, u32, u32, i32, ip, C.POINTER(u32)]
    L.tetsim_prep_ref_grab_texels.argtypes = [i32, u32, u32, ip]
    L.tetsim_prep_rest.argtypes = [fp, u32, ip, u32, dbl, fp, fp, fp]
    L.tetsim_prep_partition.argtypes = [fp, u32, ip, u32, i32, ip]
    L.tetsim_prep_partition_quality.argtypes = [ip, u32, u32, i32, ip, C.POINTER(TetSimPartQuality)]
    L.tetsim_plan_create.argtypes = [ip, u32, u32, i32, i32, ip, C.POINTER(H)]
    L.tetsim_plan_destroy.argtypes = [H]
    L.tetsim_plan_destroy.restype = None
    L.tetsim_plan_sizes.argtypes = [H, C.POINTER(TetSimPlanSizes)]
    L.tetsim_plan_arrays.argtypes = [H, ip, ip, ip]
    L.tetsim_plan_neighbour.argtypes = [H, u32, ip, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), ip]
    L.tetsim_plan_neighbour_ids.argtypes = [H, u32, ip, ip, ip]
    L.tetsim_plan_create_deep.argtypes = [ip, u32, u32, i32, i32, ip, i32, C.POINTER(H)]
    L.tetsim_plan_layers.argtypes = [H, C.POINTER(u32), C.POINTER(C.c_uint8)]
    L.tetsim_plan_neighbour_layer2.argtypes = [H, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.tetsim_plan_neighbour_layer2_ids.argtypes = [H, u32, ip, ip, ip]
    for s in SYMBOLS:
        f = getattr(L, s)
        if s not in ("tetsim_default_options", "tetsim_default_params", "tetsim_destroy", "tetsim_last_error",
                     "tetsim_plan_destroy"):
            f.restype = C.c_int
    _lib = L
    return L


def library_info():
    """{"abi", "ablation", "source_sha", "kernel_sha", "debug_env": [names set]} of the loaded libtetsim_hip.so."""
    li = TetSimLibraryInfo()
    check(lib().tetsim_library_info(C.byref(li)))
    return {"abi": li.abi, "ablation": bool(li.ablation), "source_sha": li.source_sha.decode(), "kernel_sha": li.kernel_sha.decode(),
            "debug_env": [n for i, n in enumerate(DEBUG_ENV_NAMES) if li.debug_env >> i & 1], "path": LIB_PATH}


def check(rc, handle=None):
    if rc != OK:
        msg = lib().tetsim_last_error(handle)
        raise TetSimError(rc, msg.decode("utf-8", "replace") if msg else "")
