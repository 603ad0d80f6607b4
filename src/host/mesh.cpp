// src/host/mesh.cpp -- OBJ loading and the plane/box generators of PathTrace/scene/mesh.h (scene construction, host only).
#include <PathTrace/scene/mesh.h>

#include <cmath>
#include <fstream>
#include <iterator>
#include <limits>
#include <string>

namespace {

    // A cursor over the whole file in memory.  Numbers are the longest runs of [0-9+-.eE] (floats) / [0-9+-eE] (integers);
    // whatever character ends a run is consumed with it, as a character-by-character reader would do.
    class ObjText {
      public:
        explicit ObjText(std::string text) : text(std::move(text)) {}

        bool atEnd() const { return pos >= text.size(); }
        char peek() const { return atEnd() ? static_cast<char>(-1) : text[pos]; }
        char take() { return atEnd() ? static_cast<char>(-1) : text[pos++]; }
        bool takeIf(char c) {
            if(!atEnd() && text[pos] == c) {
                pos++;
                return true;
            }
            return false;
        }
        void skipBlanks() {
            while(takeIf(' ')) {
            }
        }
        void skipLine() {
            while(!atEnd()) {
                const char c = take();
                if(c == '\r' || c == '\n') {
                    return;
                }
            }
        }
        std::string number(bool allow_point) {
            skipBlanks();
            std::string word;
            while(!atEnd()) {
                const char c = take();
                const bool part = (c >= '0' && c <= '9') || c == '-' || c == '+' || c == 'e' || c == 'E' || (allow_point && c == '.');
                if(!part) {
                    break;
                }
                word.push_back(c);
            }
            return word;
        }
        int integer() {
            try {
                return std::stoi(number(false));
            }
            catch(const std::exception &) {
                return -1;
            }
        }
        float real() {
            try {
                return std::stof(number(true));
            }
            catch(const std::exception &) {
                return std::numeric_limits<float>::quiet_NaN();
            }
        }

      private:
        std::string text;
        size_t pos = 0;
    };

    struct Corner {
        size_t face;
        int slot; // 0, 1, 2 = a, b, c
    };

} // namespace

namespace io {

    std::vector<Triangle> loadMesh(std::basic_istream<char> &stream, mat4<float> transformation, bool cull_backface, bool smooth) {
        ObjText text{std::string(std::istreambuf_iterator<char>(stream), std::istreambuf_iterator<char>())};
        std::vector<vec3<float>> vertices;
        std::vector<std::vector<Corner>> corners_of_vertex;
        std::vector<Triangle> faces;

        while(!text.atEnd()) {
            text.skipBlanks();
            const char tag = text.take();
            if(tag == '\r' || tag == '\n') {
                continue;
            }
            if(tag == 'v' && text.takeIf(' ')) {
                const float x = text.real();
                const float y = text.real();
                const float z = text.real();
                vertices.push_back(vec3<float>(transformation * vec3<float>{x, y, z}));
                corners_of_vertex.emplace_back();
            }
            else if(tag == 'f' && text.takeIf(' ')) {
                int index[3];
                for(int &i : index) {
                    i = text.integer() - 1; // OBJ indices start at 1
                    while(text.takeIf('/')) { // texture / normal references are read and ignored
                        text.integer();
                    }
                }
                const int count = static_cast<int>(vertices.size());
                if(index[0] < 0 || index[0] >= count || index[1] < 0 || index[1] >= count || index[2] < 0 || index[2] >= count) {
                    continue;
                }
                const auto &pa = vertices[index[0]];
                const auto &pb = vertices[index[1]];
                const auto &pc = vertices[index[2]];
                // three distinct points (written so that NaN coordinates fail the test) that are not collinear
                if(!((pb - pa).getLengthSquared() > 0.0F && (pc - pa).getLengthSquared() > 0.0F && (pc - pb).getLengthSquared() > 0.0F)) {
                    continue;
                }
                if(cross(pb - pa, pc - pa).getLengthSquared() <= 0.0F) {
                    continue;
                }
                for(int slot = 0; slot < 3; slot++) {
                    corners_of_vertex[index[slot]].push_back(Corner{faces.size(), slot});
                }
                faces.emplace_back(pa, pb, pc, cull_backface);
            }
            else {
                text.skipLine();
            }
        }

        if(smooth) {
            std::vector<vec3<float>> unit_face_normal;
            unit_face_normal.reserve(faces.size());
            for(const auto &f : faces) {
                unit_face_normal.push_back(vec3<float>(cross(f.b - f.a, f.c - f.a).normalize()));
            }
            for(const auto &corners : corners_of_vertex) {
                vec3<float> sum{};
                for(const Corner &corner : corners) {
                    sum = sum + unit_face_normal[corner.face];
                }
                if(sum.getLengthSquared() <= 0.0F) {
                    continue;
                }
                const vec3<float> shared = sum.normalize();
                for(const Corner &corner : corners) {
                    Triangle &f = faces[corner.face];
                    (corner.slot == 0 ? f.normal_a : corner.slot == 1 ? f.normal_b : f.normal_c) = shared;
                }
            }
        }
        return faces;
    }

    std::vector<Triangle> loadMesh(const std::filesystem::path &path, mat4<float> transformation, bool cull_backface, bool smooth) {
        std::ifstream stream(path, std::ios_base::in | std::ios_base::binary);
        return loadMesh(stream, transformation, cull_backface, smooth);
    }

} // namespace io

std::vector<Triangle> makePlane(vec3<float> a, vec3<float> b, bool cull_backface) {
    const float tolerance = 1E-4F;
    int flat_axis = -1; // the LAST axis in which the corners coincide
    int flat_count = 0;
    for(int axis = 0; axis < 3; axis++) {
        if(std::abs(a[axis] - b[axis]) < tolerance) {
            flat_axis = axis;
            flat_count++;
        }
    }
    if(flat_count != 1) {
        return {};
    }
    const int swap_axis = flat_axis == 0 ? 1 : 0;
    vec3<float> corner_ab = a;
    vec3<float> corner_ba = b;
    corner_ab[swap_axis] = b[swap_axis];
    corner_ba[swap_axis] = a[swap_axis];
    std::vector<Triangle> triangles;
    triangles.reserve(2);
    triangles.emplace_back(a, corner_ab, b, cull_backface);
    triangles.emplace_back(b, corner_ba, a, cull_backface);
    return triangles;
}

std::vector<Triangle> makeBox(vec3<float> a, vec3<float> b, bool cull_backface) {
    const float tolerance = 1E-4F;
    for(int axis = 0; axis < 3; axis++) {
        if(std::abs(a[axis] - b[axis]) < tolerance) {
            return {};
        }
    }
    std::vector<Triangle> triangles;
    triangles.reserve(12);
    for(int axis = 0; axis < 3; axis++) {
        for(const float level : {a[axis], b[axis]}) { // the two faces perpendicular to `axis`
            vec3<float> lo = a;
            vec3<float> hi = b;
            lo[axis] = level;
            hi[axis] = level;
            const auto face = makePlane(lo, hi, cull_backface);
            triangles.insert(triangles.end(), face.begin(), face.end());
        }
    }
    return triangles;
}
