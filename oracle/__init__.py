"""CPU oracle (test infrastructure only) -- see oracle/tetsim_oracle.c.

Importable ONLY from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

    tetsim_oracle.c   C restatement of both reference solvers (section A pinned bit-exactly to Softbody.js, section G
                      pinned to the reference's GLSL passes run on Mesa softpipe)
    nh_port.js        the Softbody.js algorithm in plain JavaScript (bit-exact with the same goldens): the single-thread
                      "JS CPU path" that bench.py times under node beside the GPU numbers
    glsl_ref/         headless Mesa GL ES binding + a WebGL2 context object on it: the reference's own three.js WebGLRenderer runs on them to RECORD the polar golden vectors (build container only)
"""
from .orc import OracleNH, OraclePJ, OrcParams, build_oracle, max_threads, set_threads  # noqa: F401
