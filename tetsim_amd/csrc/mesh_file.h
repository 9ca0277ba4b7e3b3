// mesh_file.h -- .tetsim binary mesh container (see mesh_file.cpp); internal C++ side of tetsim_mesh_* in include/tetsim.h.
#pragma once
#include <string>

#include "../../include/tetsim.h"

namespace tetsim {
struct MeshFile;
std::string mesh_write(const char* path, const TetSimMeshArrays& a);  // "" on success, else the reason
std::string mesh_open(const char* path, MeshFile** out);
const TetSimMeshArrays& mesh_arrays(const MeshFile* m);
void mesh_close(MeshFile* m);
}  // namespace tetsim
