// mesh_io.hpp -- mesh file readers (mesh_io.cpp)
#pragma once
#include <string>
#include <vector>
#include "../../../include/mi355rt.h"

namespace mi355rt_host {
int load_obj(const std::string& path, std::vector<mi355rt_triangle>& tris);   // Mesh::from_obj, mesh_object.rs:59-137
int load_wo3(const std::string& path, std::vector<mi355rt_triangle>& tris, bool four_index_stride = false);   // Mesh::from_wo3, mesh_object.rs:141-259
}
