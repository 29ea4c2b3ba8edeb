/*
 * rt_oracle.c -- CPU restatement of the reference's ray-scene query path.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Plain C, single-threaded,
 * compiled with -ffp-contract=off so every float operation rounds exactly as in
 * the reference's x86-64 build (no FMA; Makefile:3 of the reference).
 *
 * Citations are paths relative to the reference checkout.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ util */

static char g_err[512];
const char* orc_last_error(void) { return g_err; }
static void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static const float EPS = 1e-9f;        /* src/util.h:9  */
#define INF ((float)INFINITY)          /* src/util.h:10 */

typedef struct { float x, y, z; } v3;

/* src/vec3.h:23-31 */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mulf(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* src/vec3.h:25: division is guarded; |a| <= eps yields the zero vector */
static inline v3 v_divf(v3 a, float s) { return fabsf(s) > EPS ? V(a.x / s, a.y / s, a.z / s) : V(0, 0, 0); }
/* src/vec3.h:32 */
static inline int v_eq(v3 a, v3 b) { return fabsf(a.x - b.x) < EPS && fabsf(a.y - b.y) < EPS && fabsf(a.z - b.z) < EPS; }
/* src/vec3.h:43 */
static inline float v_len(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* src/vec3.h:45 */
static inline v3 v_norm(v3 a) { return v_divf(a, v_len(a)); }
/* src/vec3.h:54-64 */
static inline float v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v_cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 v_abs(v3 a) { return V(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline float v_get(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* std::max(a,b) = (a<b)?b:a ; std::min(a,b) = (b<a)?b:a -- NaN behaviour matters (collision_detection.h:82-83) */
static inline float std_max(float a, float b) { return (a < b) ? b : a; }
static inline float std_min(float a, float b) { return (b < a) ? b : a; }
/* src/util.h:13-16 */
static inline float clampf(float v, float lo, float hi) { return std_max(std_min(v, hi), lo); }
/* src/color.h:17-19 */
static inline v3 clamp_color(v3 c) { return V(clampf(c.x, 0.0f, 1.0f), clampf(c.y, 0.0f, 1.0f), clampf(c.z, 0.0f, 1.0f)); }

/* ------------------------------------------------------------------ scene */

typedef struct {
    v3 pos[3];
    v3 norm[3];
    int32_t material_id;
    v3 normal; /* face normal, src/triangle.h:28-32 */
} tri_t;

typedef struct {
    v3 ambient, diffuse, specular, transmittance, emission;
    float shininess, ior, dissolve;
} mat_t;

struct orc_scene {
    uint64_t n_tri;
    tri_t* tri;
    uint32_t n_mat;
    mat_t* mat;
    uint32_t n_light;
    uint32_t* light; /* indices of emissive triangles in file order (obj_loader.cc:188-191) */
    int has_normals;
};

/* src/triangle.h:28-32 */
static v3 tri_face_normal(const tri_t* t) {
    v3 u = v_sub(t->pos[1], t->pos[0]);
    v3 v = v_sub(t->pos[2], t->pos[1]);
    return v_norm(v_cross(u, v));
}
/* src/triangle.h:34-40 */
static float tri_area(const tri_t* t) {
    v3 u = v_sub(t->pos[1], t->pos[0]);
    v3 v = v_sub(t->pos[2], t->pos[1]);
    v3 c = v_cross(u, v);
    float m = v_len(c);
    return (float)(0.5 * m);
}
/* src/vertex.h:12 */
static int vert_eq(const tri_t* a, int i, const tri_t* b, int j) { return v_eq(a->pos[i], b->pos[j]) && v_eq(a->norm[i], b->norm[j]); }
/* src/triangle.h:15-19 */
static int tri_eq(const tri_t* a, const tri_t* b) {
    return (vert_eq(a, 0, b, 0) && vert_eq(a, 1, b, 1) && vert_eq(a, 2, b, 2)) ||
           (vert_eq(a, 0, b, 1) && vert_eq(a, 1, b, 2) && vert_eq(a, 2, b, 0)) ||
           (vert_eq(a, 0, b, 2) && vert_eq(a, 1, b, 0) && vert_eq(a, 2, b, 1));
}

static void scene_finish(orc_scene* s) {
    /* face normals (obj_loader.cc:185) and light list (obj_loader.cc:188-191) */
    s->n_light = 0;
    s->light = (uint32_t*)malloc(sizeof(uint32_t) * (s->n_tri ? s->n_tri : 1));
    for (uint64_t i = 0; i < s->n_tri; ++i) {
        tri_t* t = &s->tri[i];
        t->normal = tri_face_normal(t);
        if (t->material_id >= 0 && (uint32_t)t->material_id < s->n_mat) {
            const mat_t* m = &s->mat[t->material_id];
            if (m->emission.x != 0.0 || m->emission.y != 0.0 || m->emission.z != 0.0) s->light[s->n_light++] = (uint32_t)i;
        }
    }
}

orc_scene* orc_scene_from_arrays(uint64_t n_tri, const float* pos9, const float* norm9, const int32_t* matid,
                                 uint32_t n_mat, const float* materials18) {
    orc_scene* s = (orc_scene*)calloc(1, sizeof *s);
    s->n_tri = n_tri;
    s->tri = (tri_t*)calloc(n_tri ? n_tri : 1, sizeof(tri_t));
    s->n_mat = n_mat;
    s->mat = (mat_t*)calloc(n_mat ? n_mat : 1, sizeof(mat_t));
    for (uint32_t m = 0; m < n_mat; ++m) {
        const float* f = materials18 + (size_t)m * ORC_MAT_FLOATS;
        mat_t* d = &s->mat[m];
        d->ambient = V(f[0], f[1], f[2]);
        d->diffuse = V(f[3], f[4], f[5]);
        d->specular = V(f[6], f[7], f[8]);
        d->transmittance = V(f[9], f[10], f[11]);
        d->emission = V(f[12], f[13], f[14]);
        d->shininess = f[15];
        d->ior = f[16];
        d->dissolve = f[17];
    }
    s->has_normals = norm9 != NULL;
    for (uint64_t i = 0; i < n_tri; ++i) {
        tri_t* t = &s->tri[i];
        for (int v = 0; v < 3; ++v) {
            const float* p = pos9 + i * 9 + v * 3;
            t->pos[v] = V(p[0], p[1], p[2]);
            if (norm9) {
                const float* q = norm9 + i * 9 + v * 3;
                t->norm[v] = V(q[0], q[1], q[2]);
            }
        }
        t->material_id = matid ? matid[i] : 0;
    }
    scene_finish(s);
    return s;
}

void orc_scene_free(orc_scene* s) {
    if (!s) return;
    free(s->tri);
    free(s->mat);
    free(s->light);
    free(s);
}
uint64_t orc_scene_num_triangles(const orc_scene* s) { return s->n_tri; }
uint32_t orc_scene_num_materials(const orc_scene* s) { return s->n_mat; }
uint32_t orc_scene_num_lights(const orc_scene* s) { return s->n_light; }
int orc_scene_has_normals(const orc_scene* s) { return s->has_normals; }

void orc_scene_export(const orc_scene* s, float* pos9, float* norm9, int32_t* matid, float* materials18,
                      float* face_normal3, int32_t* light_tri) {
    for (uint64_t i = 0; i < s->n_tri; ++i) {
        const tri_t* t = &s->tri[i];
        for (int v = 0; v < 3; ++v) {
            if (pos9) { pos9[i * 9 + v * 3 + 0] = t->pos[v].x; pos9[i * 9 + v * 3 + 1] = t->pos[v].y; pos9[i * 9 + v * 3 + 2] = t->pos[v].z; }
            if (norm9) { norm9[i * 9 + v * 3 + 0] = t->norm[v].x; norm9[i * 9 + v * 3 + 1] = t->norm[v].y; norm9[i * 9 + v * 3 + 2] = t->norm[v].z; }
        }
        if (matid) matid[i] = t->material_id;
        if (face_normal3) { face_normal3[i * 3] = t->normal.x; face_normal3[i * 3 + 1] = t->normal.y; face_normal3[i * 3 + 2] = t->normal.z; }
    }
    if (materials18)
        for (uint32_t m = 0; m < s->n_mat; ++m) {
            float* f = materials18 + (size_t)m * ORC_MAT_FLOATS;
            const mat_t* d = &s->mat[m];
            f[0] = d->ambient.x; f[1] = d->ambient.y; f[2] = d->ambient.z;
            f[3] = d->diffuse.x; f[4] = d->diffuse.y; f[5] = d->diffuse.z;
            f[6] = d->specular.x; f[7] = d->specular.y; f[8] = d->specular.z;
            f[9] = d->transmittance.x; f[10] = d->transmittance.y; f[11] = d->transmittance.z;
            f[12] = d->emission.x; f[13] = d->emission.y; f[14] = d->emission.z;
            f[15] = d->shininess; f[16] = d->ior; f[17] = d->dissolve;
        }
    if (light_tri)
        for (uint32_t l = 0; l < s->n_light; ++l) light_tri[l] = (int32_t)s->light[l];
}

/* ---- OBJ / MTL text loader.
 * The reference delegates to tinyobjloader (include/tiny_obj_loader.h, vendored third party, v2.0.0 API).
 * What matters for the query path is restated here: faces are emitted in file order
 * (tiny_obj_loader.h:1494-1497, 2809-2848), quads are split along the shorter diagonal
 * (tiny_obj_loader.h:1562-1605), indices may be negative/relative, materials are numbered in
 * `newmtl` order, and unset material fields default to 0 except shininess = ior = dissolve = 1
 * (tiny_obj_loader.h:1410-1420).  Coordinates go through strtod then a cast to float; on both shipped
 * scenes this equals tinyobj's own decimal parser (tiny_obj_loader.h:897-1038), which the image
 * known-answer tests confirm end to end. */

typedef struct { char name[128]; } matname_t;

static float parse_f(char** p) {
    char* e;
    while (**p == ' ' || **p == '\t') ++*p;
    double d = strtod(*p, &e);
    if (e == *p) return 0.0f;
    *p = e;
    return (float)d;
}

static int load_mtl(const char* path, orc_scene* s, matname_t** names) {
    FILE* f = fopen(path, "r");
    if (!f) { set_err("cannot open mtl %s", path); return -1; }
    char line[1024];
    uint32_t cap = 16;
    s->mat = (mat_t*)calloc(cap, sizeof(mat_t));
    *names = (matname_t*)calloc(cap, sizeof(matname_t));
    s->n_mat = 0;
    mat_t* cur = NULL;
    while (fgets(line, sizeof line, f)) {
        char* p = line;
        while (*p == ' ' || *p == '\t') ++p;
        char* h = strchr(p, '#');
        if (h) *h = 0;
        size_t n = strlen(p);
        while (n && (p[n - 1] == '\n' || p[n - 1] == '\r' || p[n - 1] == ' ' || p[n - 1] == '\t')) p[--n] = 0;
        if (!n) continue;
        if (!strncmp(p, "newmtl", 6) && isspace((unsigned char)p[6])) {
            if (s->n_mat == cap) {
                cap *= 2;
                s->mat = (mat_t*)realloc(s->mat, cap * sizeof(mat_t));
                *names = (matname_t*)realloc(*names, cap * sizeof(matname_t));
            }
            cur = &s->mat[s->n_mat];
            memset(cur, 0, sizeof *cur);
            cur->dissolve = 1.0f; cur->shininess = 1.0f; cur->ior = 1.0f;
            char* q = p + 7;
            while (*q == ' ' || *q == '\t') ++q;
            snprintf((*names)[s->n_mat].name, sizeof((*names)[0].name), "%s", q);
            s->n_mat++;
            continue;
        }
        if (!cur) continue;
        char* q = p + 2;
        if (p[0] == 'K' && p[1] == 'a' && isspace((unsigned char)p[2])) { cur->ambient.x = parse_f(&q); cur->ambient.y = parse_f(&q); cur->ambient.z = parse_f(&q); }
        else if (p[0] == 'K' && p[1] == 'd' && isspace((unsigned char)p[2])) { cur->diffuse.x = parse_f(&q); cur->diffuse.y = parse_f(&q); cur->diffuse.z = parse_f(&q); }
        else if (p[0] == 'K' && p[1] == 's' && isspace((unsigned char)p[2])) { cur->specular.x = parse_f(&q); cur->specular.y = parse_f(&q); cur->specular.z = parse_f(&q); }
        else if (((p[0] == 'K' && p[1] == 't') || (p[0] == 'T' && p[1] == 'f')) && isspace((unsigned char)p[2])) { cur->transmittance.x = parse_f(&q); cur->transmittance.y = parse_f(&q); cur->transmittance.z = parse_f(&q); }
        else if (p[0] == 'K' && p[1] == 'e' && isspace((unsigned char)p[2])) { cur->emission.x = parse_f(&q); cur->emission.y = parse_f(&q); cur->emission.z = parse_f(&q); }
        else if (p[0] == 'N' && p[1] == 'i' && isspace((unsigned char)p[2])) { cur->ior = parse_f(&q); }
        else if (p[0] == 'N' && p[1] == 's' && isspace((unsigned char)p[2])) { cur->shininess = parse_f(&q); }
        else if (p[0] == 'd' && isspace((unsigned char)p[1])) { q = p + 1; cur->dissolve = parse_f(&q); }
    }
    fclose(f);
    return 0;
}

typedef struct { int v, vt, vn; } fidx_t;

static int fix_index(int idx, int n) { /* tiny_obj_loader.h fixIndex: 1-based, negative = relative */
    if (idx > 0) return idx - 1;
    if (idx < 0) return n + idx;
    return -1;
}

orc_scene* orc_scene_load_obj(const char* obj_path, const char* mtl_basepath, int triangulate) {
    FILE* f = fopen(obj_path, "r");
    if (!f) { set_err("cannot open obj %s", obj_path); return NULL; }
    orc_scene* s = (orc_scene*)calloc(1, sizeof *s);
    matname_t* names = NULL;
    size_t vcap = 1024, ncap = 1024, tcap = 1024, nv = 0, nn = 0, nt = 0, nvt = 0;
    float* vs = (float*)malloc(vcap * 3 * sizeof(float));
    float* ns = (float*)malloc(ncap * 3 * sizeof(float));
    s->tri = (tri_t*)malloc(tcap * sizeof(tri_t));
    int cur_mat = -1;
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        char* p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == '#' || *p == '\n' || *p == '\r' || !*p) continue;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* q = p + 2;
            if (nv == vcap) { vcap *= 2; vs = (float*)realloc(vs, vcap * 3 * sizeof(float)); }
            vs[nv * 3] = parse_f(&q); vs[nv * 3 + 1] = parse_f(&q); vs[nv * 3 + 2] = parse_f(&q);
            nv++;
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            char* q = p + 3;
            if (nn == ncap) { ncap *= 2; ns = (float*)realloc(ns, ncap * 3 * sizeof(float)); }
            ns[nn * 3] = parse_f(&q); ns[nn * 3 + 1] = parse_f(&q); ns[nn * 3 + 2] = parse_f(&q);
            nn++;
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            nvt++;
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            fidx_t fi[8];
            int k = 0;
            char* q = p + 2;
            for (;;) {
                while (*q == ' ' || *q == '\t' || *q == '\r') ++q;
                if (!*q || *q == '\n' || *q == '#') break;
                if (k == 8) { set_err("polygon with more than 8 vertices unsupported"); goto fail; }
                fidx_t a = {0, 0, 0};
                a.v = (int)strtol(q, &q, 10);
                if (*q == '/') {
                    ++q;
                    if (*q != '/') a.vt = (int)strtol(q, &q, 10);
                    if (*q == '/') { ++q; a.vn = (int)strtol(q, &q, 10); }
                }
                a.v = fix_index(a.v, (int)nv);
                a.vn = fix_index(a.vn, (int)nn);
                a.vt = fix_index(a.vt, (int)nvt);
                if (a.v < 0 || (size_t)a.v >= nv) { set_err("bad vertex index in face"); goto fail; }
                fi[k++] = a;
            }
            if (k < 3) continue;
            int order[2][3];
            int ntri_out = 1;
            order[0][0] = 0; order[0][1] = 1; order[0][2] = 2;
            if (k == 4 && triangulate) {
                const float* a = vs + fi[0].v * 3; const float* b = vs + fi[1].v * 3;
                const float* c = vs + fi[2].v * 3; const float* d = vs + fi[3].v * 3;
                float e02x = c[0] - a[0], e02y = c[1] - a[1], e02z = c[2] - a[2];
                float e13x = d[0] - b[0], e13y = d[1] - b[1], e13z = d[2] - b[2];
                float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
                float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                ntri_out = 2;
                if (sqr02 < sqr13) { order[0][0] = 0; order[0][1] = 1; order[0][2] = 2; order[1][0] = 0; order[1][1] = 2; order[1][2] = 3; }
                else { order[0][0] = 0; order[0][1] = 1; order[0][2] = 3; order[1][0] = 1; order[1][1] = 2; order[1][2] = 3; }
            } else if (k != 3) {
                set_err("only triangles and (triangulated) quads are supported");
                goto fail;
            }
            for (int t = 0; t < ntri_out; ++t) {
                if (nt == tcap) { tcap *= 2; s->tri = (tri_t*)realloc(s->tri, tcap * sizeof(tri_t)); }
                tri_t* T = &s->tri[nt++];
                memset(T, 0, sizeof *T);
                for (int v = 0; v < 3; ++v) {
                    fidx_t a = fi[order[t][v]];
                    T->pos[v] = V(vs[a.v * 3], vs[a.v * 3 + 1], vs[a.v * 3 + 2]);
                    if (a.vn >= 0 && (size_t)a.vn < nn) T->norm[v] = V(ns[a.vn * 3], ns[a.vn * 3 + 1], ns[a.vn * 3 + 2]);
                }
                T->material_id = cur_mat;
            }
        } else if (!strncmp(p, "usemtl", 6)) {
            char* q = p + 6;
            while (*q == ' ' || *q == '\t') ++q;
            size_t n = strlen(q);
            while (n && isspace((unsigned char)q[n - 1])) q[--n] = 0;
            cur_mat = -1;
            for (uint32_t m = 0; m < s->n_mat; ++m)
                if (!strcmp(names[m].name, q)) { cur_mat = (int)m; break; }
        } else if (!strncmp(p, "mtllib", 6) && isspace((unsigned char)p[6])) {
            char* q = p + 7;
            while (*q == ' ' || *q == '\t') ++q;
            size_t n = strlen(q);
            while (n && isspace((unsigned char)q[n - 1])) q[--n] = 0;
            char path[2048];
            snprintf(path, sizeof path, "%s%s", mtl_basepath ? mtl_basepath : "", q);
            if (load_mtl(path, s, &names)) goto fail;
        }
    }
    fclose(f);
    s->n_tri = nt;
    s->has_normals = nn > 0;
    for (uint64_t i = 0; i < s->n_tri; ++i)
        if (s->tri[i].material_id < 0 || (uint32_t)s->tri[i].material_id >= s->n_mat) {
            set_err("triangle %llu has no valid material (the reference would index out of range, obj_loader.cc:184-188)", (unsigned long long)i);
            free(vs); free(ns); free(names); orc_scene_free(s);
            return NULL;
        }
    free(vs); free(ns); free(names);
    scene_finish(s);
    return s;
fail:
    fclose(f);
    free(vs); free(ns); free(names);
    orc_scene_free(s);
    return NULL;
}

/* ------------------------------------------------------------------ primitives */

/* src/collision_detection.h:12-68 (Moeller-Trumbore; `1.0 / d` is a double divide rounded to float,
 * which equals the correctly rounded float divide). */
static inline float ray_triangle(v3 o, v3 s, const tri_t* tri, int cull) {
    v3 a = tri->pos[0], b = tri->pos[1], c = tri->pos[2];
    v3 e1 = v_sub(b, a);
    v3 e2 = v_sub(c, a);
    v3 p = v_cross(s, e2);
    float d = v_dot(e1, p);
    if (cull) {
        if (d < EPS) return INF;
    } else {
        if (fabsf(d) < EPS) return INF;
    }
    float d_inv = (float)(1.0 / (double)d);
    v3 q = v_sub(o, a);
    float u = d_inv * v_dot(q, p);
    if (u < 0.0 || u > 1.0) return INF;
    v3 w = v_cross(q, e1);
    float v = d_inv * v_dot(w, s);
    if (v < 0.0 || (u + v) > 1.0) return INF;
    return d_inv * v_dot(e2, w);
}

/* src/collision_detection.h:71-86 (slab test; no t >= 0 clipping; NaN-sensitive min/max) */
static inline int ray_aabb(v3 o, v3 dir, v3 bmin, v3 bmax) {
    float t_min = -INF, t_max = INF;
    for (int i = 0; i < 3; ++i) {
        float inv_d = 1.0f / v_get(dir, i);
        float t0 = (v_get(bmin, i) - v_get(o, i)) * inv_d;
        float t1 = (v_get(bmax, i) - v_get(o, i)) * inv_d;
        if (inv_d < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        t_min = std_max(t_min, t0);
        t_max = std_min(t_max, t1);
    }
    return t_max >= t_min;
}

static inline float min3(float a, float b, float c) { /* std::min({a,b,c}) = min_element: first smallest */
    float m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    return m;
}
static inline float max3(float a, float b, float c) { /* std::max({a,b,c}) = max_element: first largest */
    float m = a;
    if (m < b) m = b;
    if (m < c) m = c;
    return m;
}

/* src/collision_detection.h:89-138 (13-axis SAT, uses the stored normalised face normal) */
static int tri_aabb(const tri_t* t, v3 bmin, v3 bmax) {
    v3 center = v_mulf(v_add(bmin, bmax), 0.5f);
    v3 half = v_mulf(v_sub(bmax, bmin), 0.5f);
    v3 v0 = v_sub(t->pos[0], center), v1 = v_sub(t->pos[1], center), v2 = v_sub(t->pos[2], center);
    v3 e0 = v_sub(v1, v0), e1 = v_sub(v2, v1), e2 = v_sub(v0, v2);
    for (int i = 0; i < 3; ++i) {
        float mn = min3(v_get(v0, i), v_get(v1, i), v_get(v2, i));
        float mx = max3(v_get(v0, i), v_get(v1, i), v_get(v2, i));
        if (mn > v_get(half, i) || mx < -v_get(half, i)) return 0;
    }
    v3 n = t->normal;
    float p0 = v_dot(v0, n);
    float r = half.x * fabsf(n.x) + half.y * fabsf(n.y) + half.z * fabsf(n.z);
    if (fabsf(p0) > r) return 0;
    v3 axes[9] = {V(0, -e0.z, e0.y), V(0, -e1.z, e1.y), V(0, -e2.z, e2.y), V(e0.z, 0, -e0.x), V(e1.z, 0, -e1.x),
                  V(e2.z, 0, -e2.x), V(-e0.y, e0.x, 0), V(-e1.y, e1.x, 0), V(-e2.y, e2.x, 0)};
    for (int k = 0; k < 9; ++k) {
        v3 ax = axes[k];
        float q0 = v_dot(v0, ax), q1 = v_dot(v1, ax), q2 = v_dot(v2, ax);
        float rr = half.x * fabsf(ax.x) + half.y * fabsf(ax.y) + half.z * fabsf(ax.z);
        float mn = min3(q0, q1, q2), mx = max3(q0, q1, q2);
        if (mn > rr || mx < -rr) return 0;
    }
    return 1;
}

float orc_ray_triangle(const float* o, const float* d, const float* tri9, int cull) {
    tri_t t;
    memset(&t, 0, sizeof t);
    for (int v = 0; v < 3; ++v) t.pos[v] = V(tri9[v * 3], tri9[v * 3 + 1], tri9[v * 3 + 2]);
    return ray_triangle(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), &t, cull);
}
int orc_ray_aabb(const float* o, const float* d, const float* bmin, const float* bmax) {
    return ray_aabb(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), V(bmin[0], bmin[1], bmin[2]), V(bmax[0], bmax[1], bmax[2]));
}
int orc_tri_aabb(const float* tri9, const float* normal3, const float* bmin, const float* bmax) {
    tri_t t;
    memset(&t, 0, sizeof t);
    for (int v = 0; v < 3; ++v) t.pos[v] = V(tri9[v * 3], tri9[v * 3 + 1], tri9[v * 3 + 2]);
    if (normal3) t.normal = V(normal3[0], normal3[1], normal3[2]);
    else t.normal = tri_face_normal(&t);
    return tri_aabb(&t, V(bmin[0], bmin[1], bmin[2]), V(bmax[0], bmax[1], bmax[2]));
}

/* ------------------------------------------------------------------ octree */

typedef struct {
    uint8_t is_leaf;
    uint32_t depth;
    v3 bmin, bmax;
    uint64_t tri_begin; /* into pool */
    uint32_t tri_count;
    int32_t child[8];
} node_t;

struct orc_octree {
    const orc_scene* scene;
    uint32_t max_per_bb, max_depth;
    node_t* node;
    uint64_t n_node, cap_node;
    uint32_t* pool;
    uint64_t n_pool, cap_pool;
    int empty;
};

static uint64_t pool_reserve(orc_octree* t, uint64_t n) {
    if (t->n_pool + n > t->cap_pool) {
        while (t->n_pool + n > t->cap_pool) t->cap_pool = t->cap_pool ? t->cap_pool * 2 : 1024;
        t->pool = (uint32_t*)realloc(t->pool, t->cap_pool * sizeof(uint32_t));
    }
    uint64_t at = t->n_pool;
    t->n_pool += n;
    return at;
}
static uint64_t node_new(orc_octree* t) {
    if (t->n_node == t->cap_node) {
        t->cap_node = t->cap_node ? t->cap_node * 2 : 256;
        t->node = (node_t*)realloc(t->node, t->cap_node * sizeof(node_t));
    }
    node_t* n = &t->node[t->n_node];
    memset(n, 0, sizeof *n);
    for (int o = 0; o < 8; ++o) n->child[o] = -1;
    return t->n_node++;
}

/* src/ADS/Octree/octree.cc:19-171 */
orc_octree* orc_octree_build(const orc_scene* s, uint32_t max_per_bb, uint32_t max_depth) {
    orc_octree* t = (orc_octree*)calloc(1, sizeof *t);
    t->scene = s;
    t->max_per_bb = max_per_bb;
    t->max_depth = max_depth;
    if (s->n_tri == 0) { /* octree.cc:24-27: root stays null */
        t->empty = 1;
        return t;
    }
    const float aabb_eps = 1e-6f; /* octree.cc:30 */
    /* octree.cc:33-44 */
    v3 mn = s->tri[0].pos[0], mx = s->tri[0].pos[0];
    for (uint64_t i = 0; i < s->n_tri; ++i)
        for (int v = 0; v < 3; ++v) {
            v3 p = s->tri[i].pos[v];
            mn.x = p.x < mn.x ? p.x : mn.x; mn.y = p.y < mn.y ? p.y : mn.y; mn.z = p.z < mn.z ? p.z : mn.z;
            mx.x = p.x > mx.x ? p.x : mx.x; mx.y = p.y > mx.y ? p.y : mx.y; mx.z = p.z > mx.z ? p.z : mx.z;
        }
    /* octree.cc:47-54 */
    uint64_t root = node_new(t);
    t->node[root].bmin = v_sub(mn, v_abs(v_mul(mn, V(aabb_eps, aabb_eps, aabb_eps))));
    t->node[root].bmax = v_add(mx, v_abs(v_mul(mx, V(aabb_eps, aabb_eps, aabb_eps))));
    t->node[root].depth = 0;
    t->node[root].tri_begin = pool_reserve(t, s->n_tri);
    t->node[root].tri_count = (uint32_t)s->n_tri;
    for (uint64_t i = 0; i < s->n_tri; ++i) t->pool[t->node[root].tri_begin + i] = (uint32_t)i;
    if (s->n_tri <= max_per_bb) { /* octree.cc:56-63 */
        t->node[root].is_leaf = 1;
        return t;
    }
    /* BFS queue of node indices (octree.cc:65-165).  Nodes are appended in creation order, and a node is
     * only ever enqueued after creation, so a plain array with a moving head is the queue. */
    uint64_t* queue = (uint64_t*)malloc(sizeof(uint64_t) * 1024);
    uint64_t qcap = 1024, qh = 0, qt = 0;
    queue[qt++] = root;
    uint32_t* tmp[8];
    uint32_t tmp_n[8];
    uint64_t tmp_cap = 0;
    for (int o = 0; o < 8; ++o) tmp[o] = NULL;
    uint8_t* seen = (uint8_t*)calloc(s->n_tri, 1);
    while (qh < qt) {
        uint64_t ci = queue[qh++];
        node_t cur = t->node[ci];
        if (cur.depth == max_depth) { /* octree.cc:71-74 */
            t->node[ci].is_leaf = 1;
            continue;
        }
        /* octree.cc:81-83 */
        v3 half = v_divf(v_abs(v_sub(cur.bmax, cur.bmin)), 2.0f);
        v3 min_eps = v_mulf(v_abs(cur.bmin), aabb_eps);
        v3 max_eps = v_mulf(v_abs(cur.bmax), aabb_eps);
        if (cur.tri_count > tmp_cap) {
            tmp_cap = cur.tri_count;
            for (int o = 0; o < 8; ++o) tmp[o] = (uint32_t*)realloc(tmp[o], tmp_cap * sizeof(uint32_t));
        }
        int split = 0;
        v3 cmin[8], cmax[8];
        uint64_t assigned = 0;
        for (uint32_t k = 0; k < cur.tri_count; ++k) seen[t->pool[cur.tri_begin + k]] = 0;
        for (unsigned o = 0; o < 8; ++o) { /* octree.cc:87-117 */
            v3 omin = V(cur.bmin.x + (half.x * (float)((o & 4) >> 2)), cur.bmin.y + (half.y * (float)((o & 2) >> 1)),
                        cur.bmin.z + (half.z * (float)(o & 1)));
            v3 omax = v_add(omin, half);
            cmin[o] = v_sub(omin, min_eps);
            cmax[o] = v_add(omax, max_eps);
            tmp_n[o] = 0;
            for (uint32_t k = 0; k < cur.tri_count; ++k) {
                uint32_t ti = t->pool[cur.tri_begin + k];
                if (tri_aabb(&s->tri[ti], cmin[o], cmax[o])) {
                    tmp[o][tmp_n[o]++] = ti;
                    if (!seen[ti]) { seen[ti] = 1; assigned++; }
                }
            }
            if (cur.tri_count != tmp_n[o]) split = 1;
        }
        if (!split) { /* octree.cc:119-122 */
            t->node[ci].is_leaf = 1;
            continue;
        }
        if (assigned != cur.tri_count) { /* octree.cc:124-152: the reference prints a diagnostic and exit(1)s */
            set_err("octree build: %llu of %u parent triangles were assigned to no octant",
                    (unsigned long long)(cur.tri_count - assigned), cur.tri_count);
            free(queue); free(seen);
            for (int o = 0; o < 8; ++o) free(tmp[o]);
            orc_octree_free(t);
            return NULL;
        }
        for (unsigned o = 0; o < 8; ++o) { /* octree.cc:112-116,154-164 */
            if (tmp_n[o] == 0) continue;
            uint64_t ni = node_new(t);
            node_t* nn = &t->node[ni];
            nn->depth = cur.depth + 1;
            nn->bmin = cmin[o];
            nn->bmax = cmax[o];
            nn->tri_begin = pool_reserve(t, tmp_n[o]);
            nn = &t->node[ni];
            nn->tri_count = tmp_n[o];
            memcpy(t->pool + nn->tri_begin, tmp[o], tmp_n[o] * sizeof(uint32_t));
            t->node[ci].child[o] = (int32_t)ni;
            if (tmp_n[o] > max_per_bb) {
                nn->is_leaf = 0;
                if (qt == qcap) { qcap *= 2; queue = (uint64_t*)realloc(queue, qcap * sizeof(uint64_t)); }
                queue[qt++] = ni;
            } else {
                nn->is_leaf = 1;
            }
        }
    }
    free(queue);
    free(seen);
    for (int o = 0; o < 8; ++o) free(tmp[o]);
    return t;
}

void orc_octree_free(orc_octree* t) {
    if (!t) return;
    free(t->node);
    free(t->pool);
    free(t);
}

/* src/ADS/Octree/octree.cc:226-266 (leaf statistics count only leaves that are children of internal nodes) */
void orc_octree_stats(const orc_octree* t, orc_tree_stats* st) {
    memset(st, 0, sizeof *st);
    if (t->empty) return;
    long long total_depth = 0, total_tris = 0;
    for (uint64_t i = 0; i < t->n_node; ++i) {
        const node_t* n = &t->node[i];
        st->nodes++;
        if (n->is_leaf) { st->leaf_refs += n->tri_count; continue; }
        for (int o = 0; o < 8; ++o) {
            if (n->child[o] < 0) continue;
            const node_t* c = &t->node[n->child[o]];
            if (c->is_leaf) {
                st->leaves++;
                if (c->depth > st->max_depth) st->max_depth = c->depth;
                if (c->tri_count > st->max_tris_in_leaf) st->max_tris_in_leaf = c->tri_count;
                total_depth += c->depth;
                total_tris += c->tri_count;
            }
        }
    }
    st->avg_leaf_depth = (float)total_depth / st->leaves;
    st->avg_tris_in_leaf = (float)total_tris / st->leaves;
}
uint64_t orc_octree_num_nodes(const orc_octree* t) { return t->n_node; }
uint64_t orc_octree_pool_size(const orc_octree* t) { return t->n_pool; }
void orc_octree_export_pool(const orc_octree* t, uint32_t* pool) { memcpy(pool, t->pool, t->n_pool * sizeof(uint32_t)); }
void orc_octree_export(const orc_octree* t, uint8_t* is_leaf, uint32_t* depth, float* bmin3, float* bmax3, int32_t* child8,
                       uint64_t* tri_begin, uint32_t* tri_count) {
    for (uint64_t i = 0; i < t->n_node; ++i) {
        const node_t* n = &t->node[i];
        if (is_leaf) is_leaf[i] = n->is_leaf;
        if (depth) depth[i] = n->depth;
        if (bmin3) { bmin3[i * 3] = n->bmin.x; bmin3[i * 3 + 1] = n->bmin.y; bmin3[i * 3 + 2] = n->bmin.z; }
        if (bmax3) { bmax3[i * 3] = n->bmax.x; bmax3[i * 3 + 1] = n->bmax.y; bmax3[i * 3 + 2] = n->bmax.z; }
        if (child8) for (int o = 0; o < 8; ++o) child8[i * 8 + o] = n->child[o];
        if (tri_begin) tri_begin[i] = n->tri_begin;
        if (tri_count) tri_count[i] = n->tri_count;
    }
}

/* ---- Search (src/ADS/Octree/octree.cc:173-207).  The candidate list is produced into a growable
 * scratch vector owned by the caller context (one per thread of use; the oracle is single-threaded). */
typedef struct {
    uint32_t* idx;
    uint64_t n, cap;
    uint64_t* q;
    uint64_t qcap;
} cand_t;

static void cand_push(cand_t* c, const uint32_t* src, uint32_t n) {
    if (c->n + n > c->cap) {
        while (c->n + n > c->cap) c->cap = c->cap ? c->cap * 2 : 256;
        c->idx = (uint32_t*)realloc(c->idx, c->cap * sizeof(uint32_t));
    }
    memcpy(c->idx + c->n, src, n * sizeof(uint32_t));
    c->n += n;
}

static void search(const orc_octree* t, int ads, v3 o, v3 d, cand_t* c, orc_stats* st) {
    c->n = 0;
    if (st) st->search_calls++;
    if (ads == ORC_ADS_NONE) { /* src/ADS/DummyAds/dummy_ads.h:17-21 */
        const orc_scene* s = t->scene;
        if (c->cap < s->n_tri) { c->cap = s->n_tri; c->idx = (uint32_t*)realloc(c->idx, c->cap * sizeof(uint32_t)); }
        for (uint64_t i = 0; i < s->n_tri; ++i) c->idx[i] = (uint32_t)i;
        c->n = s->n_tri;
        if (st) st->search_returned += c->n;
        return;
    }
    if (t->empty) return; /* the reference would dereference a null root here (octree.cc:24-27,177) */
    if (c->qcap < t->n_node) { c->qcap = t->n_node; c->q = (uint64_t*)realloc(c->q, c->qcap * sizeof(uint64_t)); }
    uint64_t qh = 0, qt = 0;
    c->q[qt++] = 0;
    while (qh < qt) {
        const node_t* n = &t->node[c->q[qh++]];
        if (st) st->search_nodes++;
        if (n->is_leaf) {
            if (st) st->search_leaves++;
            cand_push(c, t->pool + n->tri_begin, n->tri_count);
            continue;
        }
        for (int k = 0; k < 8; ++k) {
            if (n->child[k] < 0) continue;
            const node_t* ch = &t->node[n->child[k]];
            if (ch->tri_count == 0) continue;
            if (st) st->slab_tests++;
            if (ray_aabb(o, d, ch->bmin, ch->bmax)) c->q[qt++] = (uint64_t)n->child[k];
        }
    }
    if (st) st->search_returned += c->n;
}

uint64_t orc_search(const orc_octree* t, const float* o, const float* d, uint32_t* out, uint64_t cap, orc_stats* st) {
    cand_t c;
    memset(&c, 0, sizeof c);
    search(t, ORC_ADS_OCTREE, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), &c, st);
    for (uint64_t i = 0; i < c.n && i < cap; ++i) out[i] = c.idx[i];
    uint64_t n = c.n;
    free(c.idx);
    free(c.q);
    return n;
}

/* closest-hit loop, src/Renderer/renderer.cc:116-142.  Returns index INTO THE CANDIDATE LIST (or -1). */
static int64_t closest_hit(const orc_scene* s, const cand_t* c, v3 o, v3 d, int cull, float* t_out, orc_stats* st) {
    float t_pixel = INF;
    int64_t hit = -1;
    for (uint64_t i = 0; i < c->n; ++i) {
        float t = ray_triangle(o, d, &s->tri[c->idx[i]], cull);
        if (t == INF || t < EPS) continue;
        if (t < t_pixel) { t_pixel = t; hit = (int64_t)i; }
    }
    if (st) st->tri_tests += c->n;
    *t_out = t_pixel;
    return hit;
}

void orc_trace_closest(const orc_scene* s, const orc_octree* t, uint64_t n, const float* orig, const float* dir, int cull,
                       float* out_t, int32_t* out_tri, orc_stats* st) {
    cand_t c;
    memset(&c, 0, sizeof c);
    for (uint64_t i = 0; i < n; ++i) {
        v3 o = V(orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2]);
        v3 d = V(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]);
        search(t, ORC_ADS_OCTREE, o, d, &c, st);
        float tt;
        int64_t h = closest_hit(s, &c, o, d, cull, &tt, st);
        out_t[i] = tt;
        out_tri[i] = (h >= 0 && tt != INF) ? (int32_t)c.idx[h] : -1;
    }
    free(c.idx);
    free(c.q);
}

/* src/Renderer/renderer.cc:288-313 */
static int is_shadowed(const orc_scene* s, const orc_octree* t, int ads, cand_t* c, v3 p, v3 light, int cull, orc_stats* st) {
    float dist_light = v_len(v_sub(light, p));
    v3 dir = v_norm(v_sub(light, p));
    uint64_t slab0 = st ? st->slab_tests : 0;
    if (st) st->shadow_rays++;
    search(t, ads, p, dir, c, st);
    if (st) st->slab_tests_any += st->slab_tests - slab0;
    for (uint64_t i = 0; i < c->n; ++i) {
        float sh = ray_triangle(p, dir, &s->tri[c->idx[i]], cull);
        if (sh == INF || sh < EPS) continue;
        if (sh < dist_light) {
            if (st) { st->tri_tests += i + 1; st->tri_tests_any += i + 1; }
            return 1;
        }
    }
    if (st) { st->tri_tests += c->n; st->tri_tests_any += c->n; }
    return 0;
}

void orc_trace_any(const orc_scene* s, const orc_octree* t, uint64_t n, const float* point, const float* light, int cull,
                   uint8_t* out_mask, orc_stats* st) {
    cand_t c;
    memset(&c, 0, sizeof c);
    for (uint64_t i = 0; i < n; ++i)
        out_mask[i] = (uint8_t)is_shadowed(s, t, ORC_ADS_OCTREE, &c, V(point[i * 3], point[i * 3 + 1], point[i * 3 + 2]),
                                           V(light[i * 3], light[i * 3 + 1], light[i * 3 + 2]), cull, st);
    free(c.idx);
    free(c.q);
}

/* ------------------------------------------------------------------ RNG */

/* Counter-based generator shared (bit for bit) with the device code: a chain of 32-bit avalanche
 * steps over the key fields, reduced to 31 bits so that it maps to float exactly like
 * (float)rand() / (float)RAND_MAX (src/triangle.h:43-44, renderer.cc:44-45): k / 2^31, which can be 1.0. */
static inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU;
    h ^= h >> 15; h *= 0x846ca68bU;
    h ^= h >> 16;
    return h;
}
static inline uint32_t rng_prefix(uint32_t seed, uint32_t pixel, uint32_t spp_sample, uint32_t node_id) {
    uint32_t h = mix32(seed + 0x9e3779b9U);
    h = mix32(h ^ pixel);
    h = mix32(h ^ (spp_sample + 0x85ebca6bU));
    h = mix32(h ^ (node_id + 0xc2b2ae35U));
    return h;
}
static inline uint32_t rng_finish(uint32_t prefix, uint32_t light, uint32_t sample, uint32_t dim) {
    uint32_t h = mix32(prefix ^ (light + 0x27d4eb2fU));
    h = mix32(h ^ (sample + 0x165667b1U));
    h = mix32(h ^ (dim + 0xd3a2646cU));
    return h >> 1;
}
uint32_t orc_rng_u31(uint32_t seed, uint32_t pixel, uint32_t spp_sample, uint32_t node_id, uint32_t light, uint32_t sample,
                     uint32_t dim) {
    return rng_finish(rng_prefix(seed, pixel, spp_sample, node_id), light, sample, dim);
}
static inline float u31_to_float(uint32_t k) { return (float)(int32_t)k / (float)2147483647; }

/* ------------------------------------------------------------------ renderer */

typedef struct {
    const orc_scene* s;
    const orc_octree* t;
    const orc_render_cfg* cfg;
    orc_stats* st;
    cand_t c;
    float* light_area;
    /* counter-RNG context */
    uint32_t pixel, spp_sample;
} rctx;

/* src/Renderer/renderer.cc:82-99: camera frame.  `gh = gw * (height / width)` divides ints. */
void orc_camera_frame(const orc_render_cfg* cfg, float* p00o, float* qwo, float* qho) {
    int width = cfg->width, height = cfg->height;
    v3 dir = V(cfg->cam_dir[0], cfg->cam_dir[1], cfg->cam_dir[2]);
    v3 up = V(cfg->cam_up[0], cfg->cam_up[1], cfg->cam_up[2]);
    float t = 1.0f;
    v3 b = v_cross(dir, up);
    float gw = 2 * t * tanf(cfg->fov / 2.0f);
    float gh = cfg->aspect_mode == ORC_ASPECT_TRUE ? gw * ((float)height / (float)width) : gw * (float)(height / width);
    v3 qw = v_mulf(b, gw / (float)(width - 1));
    v3 qh = v_mulf(up, gh / (float)(height - 1));
    v3 p00 = v_add(v_sub(v_mulf(dir, t), v_mulf(b, gw / 2)), v_mulf(up, gh / 2));
    p00o[0] = p00.x; p00o[1] = p00.y; p00o[2] = p00.z;
    qwo[0] = qw.x; qwo[1] = qw.y; qwo[2] = qw.z;
    qho[0] = qh.x; qho[1] = qh.y; qho[2] = qh.z;
}

/* src/Renderer/renderer.cc:101-108 */
static inline v3 primary_dir(v3 p00, v3 qw, v3 qh, int x, int y) {
    v3 pxy = v_sub(v_add(p00, v_mulf(qw, (float)x)), v_mulf(qh, (float)y));
    return v_divf(pxy, v_len(pxy));
}

void orc_generate_rays(const orc_render_cfg* cfg, float* out) {
    float a[3], b[3], c[3];
    orc_camera_frame(cfg, a, b, c);
    v3 p00 = V(a[0], a[1], a[2]), qw = V(b[0], b[1], b[2]), qh = V(c[0], c[1], c[2]);
    for (int y = 0; y < cfg->height; ++y)
        for (int x = 0; x < cfg->width; ++x) {
            v3 r = primary_dir(p00, qw, qh, x, y);
            float* o = out + ((size_t)y * cfg->width + x) * 3;
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
        }
}

static inline float draw(rctx* R, uint32_t node_id, uint32_t light, uint32_t sample, uint32_t dim) {
    if (R->cfg->rng_mode == ORC_RNG_LEGACY_RAND) return (float)rand() / (float)RAND_MAX;
    return u31_to_float(orc_rng_u31(R->cfg->seed, R->pixel, R->spp_sample, node_id, light, sample, dim));
}

/* src/triangle.h:42-52 */
static v3 rand_point_on_triangle(rctx* R, const tri_t* t, uint32_t node_id, uint32_t light, uint32_t sample) {
    float r1 = draw(R, node_id, light, sample, 0);
    float r2 = draw(R, node_id, light, sample, 1);
    float u = (r1 + r2 > 1) ? 1 - r1 : r1;
    float v = (r1 + r2 > 1) ? 1 - r2 : r2;
    v3 a = t->pos[0], b = t->pos[1], c = t->pos[2];
    return v_add(v_add(a, v_mulf(v_sub(b, a), u)), v_mulf(v_sub(c, a), v));
}

/* src/triangle.h:54-77 */
static v3 interpolate_normal(const tri_t* t, v3 p) {
    v3 v0 = v_sub(t->pos[1], t->pos[0]);
    v3 v1 = v_sub(t->pos[2], t->pos[0]);
    v3 v2 = v_sub(p, t->pos[0]);
    float d00 = v_dot(v0, v0), d01 = v_dot(v0, v1), d11 = v_dot(v1, v1), d20 = v_dot(v2, v0), d21 = v_dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    v3 n = v_add(v_add(v_mulf(t->norm[0], u), v_mulf(t->norm[1], v)), v_mulf(t->norm[2], w));
    return v_norm(n);
}

/* src/Renderer/renderer.cc:249-267 */
static v3 render_phong(const rctx* R, const mat_t* m, v3 P, v3 N, v3 light_pos, v3 I_l) {
    v3 cam = V(R->cfg->cam_pos[0], R->cfg->cam_pos[1], R->cfg->cam_pos[2]);
    v3 d_l = v_norm(v_sub(light_pos, P));
    v3 d_v = v_norm(v_sub(cam, P));
    v3 d_r = v_sub(v_mulf(v_mulf(N, 2.0f), v_dot(N, d_l)), d_l);
    v3 I_a = v_mul(I_l, V(0, 0, 0));
    v3 I_d = v_mulf(v_mul(I_l, m->diffuse), std_max(0.0f, v_dot(N, d_l)));
    v3 I_s = v_mulf(v_mul(I_l, m->specular), powf(std_max(0.0f, v_dot(d_v, d_r)), m->shininess));
    v3 I_e = m->emission;
    return v_add(v_add(v_add(I_a, I_d), I_s), I_e);
}
/* src/Renderer/renderer.cc:269-286 */
static v3 render_blinn_phong(const rctx* R, const mat_t* m, v3 P, v3 N, v3 light_pos, v3 I_l) {
    v3 cam = V(R->cfg->cam_pos[0], R->cfg->cam_pos[1], R->cfg->cam_pos[2]);
    v3 d_l = v_norm(v_sub(light_pos, P));
    v3 d_v = v_norm(v_sub(cam, P));
    v3 d_h = v_norm(v_add(d_l, d_v));
    v3 I_a = v_mul(I_l, V(0, 0, 0));
    v3 I_d = v_mulf(v_mul(I_l, m->diffuse), std_max(0.0f, v_dot(N, d_l)));
    v3 I_s = v_mulf(v_mul(I_l, m->specular), powf(std_max(0.0f, v_dot(N, d_h)), m->shininess));
    return v_add(v_add(I_a, I_d), I_s);
}

/* src/Renderer/renderer.cc:202-247 */
static v3 local_illumination(rctx* R, const tri_t* tri, const mat_t* m, v3 P, v3 N, uint32_t node_id) {
    const orc_scene* s = R->s;
    const orc_render_cfg* cfg = R->cfg;
    v3 final = V(0, 0, 0);
    for (uint32_t l = 0; l < s->n_light; ++l) {
        const tri_t* light = &s->tri[s->light[l]];
        const mat_t* lm = &s->mat[light->material_id];
        if (tri_eq(light, tri)) return lm->emission;
        float S_l = R->light_area[l];
        v3 acc = V(0, 0, 0);
        for (int smp = 0; smp < cfg->samples_per_triangle; ++smp) {
            v3 p_l = rand_point_on_triangle(R, light, node_id, l, (uint32_t)smp);
            int shadowed = is_shadowed(s, R->t, cfg->ads, &R->c, P, p_l, cfg->cull_backfaces, R->st);
            if (!shadowed) {
                v3 n_l = light->normal;
                v3 d_l = v_norm(v_sub(p_l, P));
                int sn = cfg->samples_per_triangle;
                float d = v_len(v_sub(p_l, P));
                float w = (S_l * std_max(0.0f, v_dot(n_l, v_neg(d_l)))) / ((float)sn * d * d + EPS);
                v3 I_l = v_mulf(lm->emission, w);
                if (cfg->render_type == ORC_RENDER_PHONG) acc = v_add(acc, render_phong(R, m, P, N, p_l, I_l));
                else acc = v_add(acc, render_blinn_phong(R, m, P, N, p_l, I_l));
            }
        }
        final = v_add(final, acc);
    }
    return final;
}

/* src/Renderer/renderer.cc:114-195.  node_id numbers the recursion tree (1 = camera ray, 2n = reflection
 * child, 2n+1 = refraction child) for the counter-based generator. */
static v3 ray_color(rctx* R, v3 o, v3 d, int depth, uint32_t node_id, int32_t* hit_tri, float* hit_t) {
    const orc_scene* s = R->s;
    const orc_render_cfg* cfg = R->cfg;
    v3 bg = V(cfg->background[0], cfg->background[1], cfg->background[2]);
    if (depth == 0) R->st->primary_rays++; else R->st->secondary_rays++;
    search(R->t, cfg->ads, o, d, &R->c, R->st);
    float t_pixel;
    int64_t h = closest_hit(s, &R->c, o, d, cfg->cull_backfaces, &t_pixel, R->st);
    if (t_pixel == INF) {
        if (hit_tri) *hit_tri = -1;
        if (hit_t) *hit_t = INF;
        return bg;
    }
    uint32_t ti = R->c.idx[h];
    if (hit_tri) *hit_tri = (int32_t)ti;
    if (hit_t) *hit_t = t_pixel;
    const tri_t* tri = &s->tri[ti];
    const mat_t* m = &s->mat[tri->material_id];
    v3 P = v_add(o, v_mulf(d, t_pixel));
    v3 N = tri->normal;
    if (cfg->shading_type == ORC_SHADING_SMOOTH) N = interpolate_normal(tri, P);
    v3 final = V(0, 0, 0);
    switch (cfg->render_type) {
        case ORC_RENDER_DISTANCE: { /* renderer.cc:197-200 */
            float t_max = cfg->max_distance;
            float g = 1.0f - (std_min(t_pixel, 1.5f * t_max) / (1.5f * t_max));
            final = v_add(final, V(g, g, g));
            break;
        }
        case ORC_RENDER_DIFFUSION: final = v_add(final, m->diffuse); break;
        default: final = v_add(final, local_illumination(R, tri, m, P, N, node_id)); break;
    }
    final = clamp_color(final);
    if (v_eq(final, V(1.0f, 1.0f, 1.0f))) return final; /* renderer.cc:172-174 */
    if (depth < cfg->max_depth) {
        if (!v_eq(m->specular, V(0, 0, 0))) { /* renderer.cc:179-183, 315-321 */
            v3 d_v = v_neg(d);
            v3 d_r = v_sub(v_mulf(v_mulf(N, 2.0f), v_dot(N, d_v)), d_v);
            v3 I_R = ray_color(R, P, v_norm(d_r), depth + 1, node_id * 2, NULL, NULL);
            final = v_add(final, v_mul(I_R, m->specular));
        }
        if (!v_eq(m->transmittance, V(0, 0, 0))) { /* renderer.cc:185-191, 323-335 */
            v3 d_v = v_neg(d);
            float n1 = 1.0f, n2 = m->ior;
            float n = n1 / n2;
            float ndotv = v_dot(N, d_v);
            /* renderer.cc:330: the unqualified `sqrt` resolves to ::sqrt(double) inside namespace raytracer, so
             * the float radicand is widened, the root stays double, `n * ndotv - root` is a double subtraction
             * and only the difference is rounded to float (by Vec3::operator*(float)). */
            double root = sqrt((double)(1 - n * n * (1 - ndotv * ndotv)));
            float coef = (float)((double)(n * ndotv) - root);
            v3 tt = v_add(v_mulf(d_v, -n), v_mulf(N, coef));
            v3 rd = V(0, 0, 0);
            if (v_len(tt) > EPS) rd = v_norm(tt);
            if (!v_eq(rd, V(0, 0, 0))) {
                v3 I_T = ray_color(R, P, rd, depth + 1, node_id * 2 + 1, NULL, NULL);
                final = v_add(final, v_mul(I_T, m->transmittance));
            }
        }
    }
    return final;
}

/* src/Renderer/renderer.cc:25-52 */
int orc_render(const orc_scene* s, const orc_octree* t, const orc_render_cfg* cfg, float* out_rgb, int32_t* out_hit_tri,
               float* out_hit_t, orc_stats* stats) {
    orc_stats local;
    memset(&local, 0, sizeof local);
    rctx R;
    memset(&R, 0, sizeof R);
    R.s = s; R.t = t; R.cfg = cfg; R.st = stats ? stats : &local;
    if (cfg->samples_per_pixel < 1) { set_err("samples_per_pixel must be >= 1 (renderer.cc:43 would wrap)"); return -1; }
    R.light_area = (float*)malloc(sizeof(float) * (s->n_light ? s->n_light : 1));
    for (uint32_t l = 0; l < s->n_light; ++l) R.light_area[l] = tri_area(&s->tri[s->light[l]]);
    if (cfg->rng_mode == ORC_RNG_LEGACY_RAND) srand(cfg->seed); /* src/main.cc:23-24 */
    float a[3], b[3], c[3];
    orc_camera_frame(cfg, a, b, c);
    v3 p00 = V(a[0], a[1], a[2]), qw = V(b[0], b[1], b[2]), qh = V(c[0], c[1], c[2]);
    v3 cam = V(cfg->cam_pos[0], cfg->cam_pos[1], cfg->cam_pos[2]);
    int y0 = 0, y1 = cfg->height;
    if (cfg->row_end > cfg->row_begin) { y0 = cfg->row_begin; y1 = cfg->row_end; }
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < cfg->width; ++x) {
            size_t pix = (size_t)y * cfg->width + x;
            R.pixel = (uint32_t)pix;
            R.spp_sample = 0;
            v3 dir = primary_dir(p00, qw, qh, x, y);
            v3 col = ray_color(&R, cam, dir, 0, 1, out_hit_tri ? &out_hit_tri[pix] : NULL, out_hit_t ? &out_hit_t[pix] : NULL);
            for (int i = 0; i < cfg->samples_per_pixel - 1; ++i) { /* renderer.cc:43-48 */
                R.spp_sample = (uint32_t)(i + 1);
                float xj = draw(&R, 0, 0, 0, 0) * 0.001f;
                float yj = draw(&R, 0, 0, 0, 1) * 0.001f;
                v3 jd = v_add(dir, V(xj, yj, 0.0f));
                col = v_add(col, ray_color(&R, cam, jd, 0, 1, NULL, NULL));
            }
            col = v_divf(col, (float)cfg->samples_per_pixel);
            col = clamp_color(col);
            out_rgb[pix * 3] = col.x; out_rgb[pix * 3 + 1] = col.y; out_rgb[pix * 3 + 2] = col.z;
        }
    free(R.light_area);
    free(R.c.idx);
    free(R.c.q);
    return 0;
}

/* src/Renderer/renderer.cc:381-390 + src/color.h:12-15 */
uint64_t orc_ppm_text(int32_t w, int32_t h, const float* rgb, char* out, uint64_t cap) {
    uint64_t n = 0;
    char buf[64];
    int k = snprintf(buf, sizeof buf, "P3\n%d %d\n255\n", w, h);
    if (out && n + (uint64_t)k <= cap) memcpy(out + n, buf, (size_t)k);
    n += (uint64_t)k;
    for (size_t i = 0; i < (size_t)w * (size_t)h; ++i) {
        k = snprintf(buf, sizeof buf, "%d %d %d\n", (int)(255.999 * rgb[i * 3]), (int)(255.999 * rgb[i * 3 + 1]),
                     (int)(255.999 * rgb[i * 3 + 2]));
        if (out && n + (uint64_t)k <= cap) memcpy(out + n, buf, (size_t)k);
        n += (uint64_t)k;
    }
    return n;
}
