"""tetsim_amd -- MI355X-native XPBD tetrahedral soft-body hot path behind TetSim's SoftBody surface.

Product code: tetsim_amd/csrc (HIP kernels + C ABI, include/tetsim.h), this thin ctypes host
(`SoftBodyHIP`) and its Node.js twin under tetsim_amd/node.  The CPU oracle lives in /oracle and is never
imported from here.
"""
from ._capi import library_info  # noqa: F401
from .lattice import make_lattice  # noqa: F401
from .softbody import (SoftBodyHIP, TetSimError, comm_info, comm_init, comm_selftest, comm_unique_id, group_p2p_connect, group_refresh_final, group_step_n, group_visual_vertex_normals,  # noqa: F401
                       halo_exchange_local, halo_p2p_probe, halo_probe, make_params, measure_copy_bandwidth, measure_stream_bandwidth, p2p_connect, p2p_export)
