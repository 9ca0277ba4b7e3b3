"""Builds the N-API shim (tetsim_napi.node) with plain g++ against the system node headers (no node-gyp)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tetsim_napi.node")


def build_addon(force=False):
    inc = None
    for cand in ("/usr/include/node", "/usr/local/include/node"):
        if os.path.exists(os.path.join(cand, "node_api.h")):
            inc = cand
            break
    if inc is None:
        raise RuntimeError("node_api.h not found (no Node.js headers on this host)")
    src = os.path.join(HERE, "tetsim_napi.cc")
    hdr = os.path.join(HERE, "..", "..", "include", "tetsim.h")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return OUT
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-DNODE_GYP_MODULE_NAME=tetsim_napi",
                           "-I" + inc, src, "-o", OUT, "-ldl"])
    return OUT


if __name__ == "__main__":
    print(build_addon(force=True))
