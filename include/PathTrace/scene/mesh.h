// PathTrace/scene/mesh.h -- triangle-mesh helpers of the PathTrace API: Wavefront OBJ loading and plane/box generators.
// Host code on the scene-construction side (src/host/mesh.cpp); not on the rendering path.
#ifndef PATHTRACE_MESH_H
#define PATHTRACE_MESH_H

#include <PathTrace/scene/object.h>

#include <filesystem>
#include <istream>
#include <memory>
#include <vector>

namespace io {

    // Reads the `v` and `f` records of an OBJ stream (other records are skipped, faces use their first three vertices, vertex
    // references may carry /vt/vn suffixes).  Every vertex is transformed first; degenerate faces are dropped.  With `smooth`
    // each vertex normal is the normalised sum of the unit normals of the faces that share the vertex.
    std::vector<Triangle> loadMesh(std::basic_istream<char> &stream, mat4<float> transformation = mat4_identity<float>, bool cull_backface = true,
                                   bool smooth = true);
    std::vector<Triangle> loadMesh(const std::filesystem::path &path, mat4<float> transformation = mat4_identity<float>, bool cull_backface = true,
                                   bool smooth = true);

} // namespace io

// Two triangles spanning the axis-aligned rectangle with opposite corners a and b (which must coincide in exactly one
// coordinate); empty otherwise.
std::vector<Triangle> makePlane(vec3<float> a, vec3<float> b, bool cull_backface = false);

// Twelve triangles of the axis-aligned box with opposite corners a and b; empty if the box is flat.
std::vector<Triangle> makeBox(vec3<float> a, vec3<float> b, bool cull_backface = false);

// appends heap copies of `extension` to `objects`
template<typename T>
void moveObjects(std::vector<std::unique_ptr<Object>> &objects, std::vector<T> &extension) {
    objects.reserve(objects.size() + extension.size());
    for(auto &item : extension) {
        objects.emplace_back(std::make_unique<T>(item));
    }
}

#endif
