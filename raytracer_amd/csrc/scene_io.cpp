// scene_io.cpp -- host-side scene input and image output around the query path.
//
//  * load_obj: what the reference's LoadScene (src/ObjLoader/obj_loader.cc:130-212 of the reference) hands to
//    Ads::Build and the renderer -- triangles in file order with per-vertex normals and a material id, and the
//    material table -- read from Wavefront OBJ/MTL text.  The reference delegates the parsing to a vendored
//    third-party header; the rules that decide the triangle soup are restated here: faces in file order, quads split
//    along the shorter diagonal ([0,1,2],[0,2,3] if |v2-v0|^2 < |v3-v1|^2, else [0,1,3],[1,2,3]), 1-based indices
//    with negative = relative to the current count, materials numbered in `newmtl` order, unset material fields
//    0 except shininess = ior = dissolve = 1.
//  * write_ppm: the P3 text the reference writes (src/Renderer/renderer.cc:381-390, src/color.h:12-15).
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "scene_io.h"

namespace rt {

namespace {

// Decimal text -> double the way the reference's vendored parser does it (include/tiny_obj_loader.h:897-1028 of the
// reference, tryParseDouble), NOT the correctly rounded strtod: integer digits accumulate as m = m * 10 + digit,
// fraction digit k adds digit * 10^-k (a table for k < 8, pow(10, -k) beyond), a decimal exponent e is applied as
// ldexp(m * 5^e, e).  The sums carry rounding errors, so a literal near a float rounding boundary can come out one ulp
// away from strtod's value -- and one ulp in a vertex can move a triangle into or out of an octant.
// Grammar: [sign] (digits | '.') ['.' digits] [('e' | 'E') [sign] digits]; returns false (value untouched) on anything else.
bool parse_decimal(const char* s, const char* end, double* out) {
    if (s >= end) return false;
    auto digit = [](char ch) { return ch >= '0' && ch <= '9'; };
    double m = 0.0;
    bool negative = false, leading_dot = false;
    const char* c = s;
    if (*c == '+' || *c == '-') {
        negative = *c == '-';
        ++c;
        leading_dot = c != end && *c == '.';
    } else if (*c == '.') {
        leading_dot = true;
    } else if (!digit(*c)) {
        return false;
    }
    if (!leading_dot) {
        int n = 0;
        for (; c != end && digit(*c); ++c, ++n) m = m * 10 + static_cast<int>(*c - '0');
        if (n == 0) return false;
    }
    int e10 = 0;
    if (c != end && *c == '.') {
        static const double kTenth[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
        ++c;
        for (int k = 1; c != end && digit(*c); ++c, ++k) m += static_cast<int>(*c - '0') * (k < 8 ? kTenth[k] : std::pow(10.0, -k));
    }
    if (c != end && (*c == 'e' || *c == 'E')) {
        ++c;
        bool e_negative = false;
        if (c != end && (*c == '+' || *c == '-')) {
            e_negative = *c == '-';
            ++c;
        } else if (c == end || !digit(*c)) {
            return false;
        }
        int n = 0;
        for (; c != end && digit(*c); ++c, ++n) {
            if (e10 > 2147483647 / 10) return false;
            e10 = e10 * 10 + static_cast<int>(*c - '0');
        }
        if (n == 0) return false;
        if (e_negative) e10 = -e10;
    }
    const double v = e10 ? std::ldexp(m * std::pow(5.0, e10), e10) : m;
    *out = negative ? -v : v;
    return true;
}

struct Cursor {
    const char* p;
    void skip_blank() {
        while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
    }
    bool at_end() const { return *p == 0 || *p == '\n' || *p == '#'; }
    float real() {  // one blank-delimited token (tiny_obj_loader.h:1030-1038: parseReal, default 0)
        skip_blank();
        const char* end = p;
        while (*end != 0 && *end != ' ' && *end != '\t' && *end != '\r' && *end != '\n') ++end;
        double v = 0.0;
        parse_decimal(p, end, &v);
        p = end;
        return static_cast<float>(v);
    }
    bool integer(long& out) {
        char* e = nullptr;
        out = std::strtol(p, &e, 10);
        if (e == p) return false;
        p = e;
        return true;
    }
    std::string rest() {
        skip_blank();
        std::string s(p);
        while (!s.empty() && std::isspace(static_cast<unsigned char>(s.back()))) s.pop_back();
        return s;
    }
};

bool keyword(const char* line, const char* kw, Cursor& c) {
    const size_t n = std::strlen(kw);
    if (std::strncmp(line, kw, n) != 0 || !(line[n] == ' ' || line[n] == '\t')) return false;
    c.p = line + n;
    return true;
}

struct Corner {
    int v = -1, vn = -1;
};

int resolve(long idx, size_t count) {  // 1-based, negative = relative to the elements seen so far
    if (idx > 0) return static_cast<int>(idx - 1);
    if (idx < 0) return static_cast<int>(static_cast<long>(count) + idx);
    return -1;
}

bool load_mtl(const std::string& path, SceneData& s, std::unordered_map<std::string, int>& ids, std::string& err) {
    std::ifstream f(path);
    if (!f) {
        err = "cannot open material file " + path;
        return false;
    }
    std::string raw;
    float* cur = nullptr;
    while (std::getline(f, raw)) {
        const size_t hash = raw.find('#');
        if (hash != std::string::npos) raw.resize(hash);
        Cursor c{raw.c_str()};
        c.skip_blank();
        const char* line = c.p;
        if (!*line) continue;
        Cursor a{line};
        if (keyword(line, "newmtl", a)) {
            ids.emplace(a.rest(), static_cast<int>(s.materials.size() / 18));
            s.materials.resize(s.materials.size() + 18, 0.0f);
            cur = &s.materials[s.materials.size() - 18];
            cur[15] = cur[16] = cur[17] = 1.0f;  // shininess, ior, dissolve
            continue;
        }
        if (!cur) continue;
        auto three = [&](int at) {
            cur[at] = a.real();
            cur[at + 1] = a.real();
            cur[at + 2] = a.real();
        };
        if (keyword(line, "Ka", a)) three(0);
        else if (keyword(line, "Kd", a)) three(3);
        else if (keyword(line, "Ks", a)) three(6);
        else if (keyword(line, "Kt", a) || keyword(line, "Tf", a)) three(9);
        else if (keyword(line, "Ke", a)) three(12);
        else if (keyword(line, "Ns", a)) cur[15] = a.real();
        else if (keyword(line, "Ni", a)) cur[16] = a.real();
        else if (keyword(line, "d", a)) cur[17] = a.real();
    }
    return true;
}

}  // namespace

bool load_obj(const std::string& obj_path, const std::string& basepath, bool triangulate, SceneData& s, std::string& err) {
    s = SceneData();
    std::ifstream f(obj_path);
    if (!f) {
        err = "Failed to load/parse .obj. (cannot open " + obj_path + ")";  // obj_loader.cc:154-156
        return false;
    }
    std::vector<float> v, vn;
    size_t n_vt = 0;
    std::unordered_map<std::string, int> mat_ids;
    int cur_mat = -1;
    std::string raw;
    size_t line_no = 0;
    while (std::getline(f, raw)) {
        ++line_no;
        Cursor c{raw.c_str()};
        c.skip_blank();
        const char* line = c.p;
        if (!*line || *line == '#') continue;
        Cursor a{line};
        if (keyword(line, "v", a)) {
            for (int k = 0; k < 3; ++k) v.push_back(a.real());
        } else if (keyword(line, "vn", a)) {
            for (int k = 0; k < 3; ++k) vn.push_back(a.real());
        } else if (keyword(line, "vt", a)) {
            ++n_vt;
        } else if (keyword(line, "f", a)) {
            std::vector<Corner> poly;
            for (;;) {
                a.skip_blank();
                if (a.at_end()) break;
                long iv = 0, ivt = 0, ivn = 0;
                if (!a.integer(iv)) {
                    err = "malformed face at line " + std::to_string(line_no);
                    return false;
                }
                if (*a.p == '/') {
                    ++a.p;
                    if (*a.p != '/') a.integer(ivt);
                    if (*a.p == '/') {
                        ++a.p;
                        a.integer(ivn);
                    }
                }
                Corner k;
                k.v = resolve(iv, v.size() / 3);
                k.vn = resolve(ivn, vn.size() / 3);
                if (k.v < 0 || static_cast<size_t>(k.v) >= v.size() / 3) {
                    err = "face with invalid vertex index at line " + std::to_string(line_no);
                    return false;
                }
                poly.push_back(k);
            }
            if (poly.size() < 3) continue;
            int order[2][3] = {{0, 1, 2}, {0, 2, 3}};
            int n_out = 1;
            if (poly.size() == 4 && triangulate) {
                const float* p0 = &v[poly[0].v * 3];
                const float* p1 = &v[poly[1].v * 3];
                const float* p2 = &v[poly[2].v * 3];
                const float* p3 = &v[poly[3].v * 3];
                const float ax = p2[0] - p0[0], ay = p2[1] - p0[1], az = p2[2] - p0[2];
                const float bx = p3[0] - p1[0], by = p3[1] - p1[1], bz = p3[2] - p1[2];
                const float d02 = ax * ax + ay * ay + az * az, d13 = bx * bx + by * by + bz * bz;
                n_out = 2;
                if (!(d02 < d13)) {
                    order[0][2] = 3;
                    order[1][0] = 1;
                    order[1][1] = 2;
                    order[1][2] = 3;
                }
            } else if (poly.size() != 3) {
                err = "only triangles and (triangulated) quads are supported (line " + std::to_string(line_no) + ")";
                return false;
            }
            if (cur_mat < 0) {
                err = "face without a valid material at line " + std::to_string(line_no) +
                      " (the reference would index its material table out of range, obj_loader.cc:184-188)";
                return false;
            }
            for (int t = 0; t < n_out; ++t) {
                for (int k = 0; k < 3; ++k) {
                    const Corner& cn = poly[order[t][k]];
                    for (int x = 0; x < 3; ++x) s.positions.push_back(v[cn.v * 3 + x]);
                    const bool has = cn.vn >= 0 && static_cast<size_t>(cn.vn) < vn.size() / 3;
                    for (int x = 0; x < 3; ++x) s.normals.push_back(has ? vn[cn.vn * 3 + x] : 0.0f);
                }
                s.material_ids.push_back(cur_mat);
            }
        } else if (keyword(line, "usemtl", a)) {
            const auto it = mat_ids.find(a.rest());
            cur_mat = it == mat_ids.end() ? -1 : it->second;
        } else if (keyword(line, "mtllib", a)) {
            if (!load_mtl(basepath + a.rest(), s, mat_ids, err)) return false;
        }
    }
    s.has_normals = !vn.empty();
    (void)n_vt;
    return true;
}

bool write_ppm(const std::string& path, int width, int height, const float* rgb, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) {
        err = "cannot open " + path + " for writing";
        return false;
    }
    std::fprintf(f, "P3\n%d %d\n255\n", width, height);
    const size_t n = static_cast<size_t>(width) * static_cast<size_t>(height);
    for (size_t i = 0; i < n; ++i)  // color.h:13-14: static_cast<int>(255.999 * c), the product taken in double
        std::fprintf(f, "%d %d %d\n", static_cast<int>(255.999 * rgb[i * 3]), static_cast<int>(255.999 * rgb[i * 3 + 1]),
                     static_cast<int>(255.999 * rgb[i * 3 + 2]));
    if (std::fclose(f) != 0) {
        err = "error while writing " + path;
        return false;
    }
    return true;
}

}  // namespace rt
