// Tangent space per triangle corner after Morten S. Mikkelsen's "mikktspace" (the algorithm the reference runs through the crate
// bevy_mikktspace 0.16.1, Cargo.lock:186-187, from gltf/buffers/tangents.rs:268-364: generate_tangents = genTangSpace with an
// angular threshold of 180 degrees, triangles only).  The crate is not in /root/reference; this restates the published algorithm:
//
//   1. corners with identical position, normal and texture coordinate are welded to one vertex id;
//   2. triangles with two equal positions are degenerate: they take no part and inherit a neighbour's result at the end;
//   3. per triangle: the first-order derivatives dP/ds (vOs) and dP/dt (vOt) of the UV map, normalised and flipped for mirrored
//      UV maps, their magnitudes, the orientation flag (signed UV area > 0), and "group with any" for a zero UV area;
//   4. neighbours: two triangles that traverse a shared edge (same two welded ids) in opposite directions;
//   5. groups ("4 rule"): around each vertex id, the triangles connected through such edges that have the same orientation — a
//      mirrored seam, a UV-chart boundary (different welded ids) or an inconsistent winding splits the vertex into several groups;
//   6. per group and corner: the sub-group of triangles whose projected derivatives lie within the angular threshold of the
//      corner's own (with 180 degrees: cos > -1), evaluated as the angle-weighted sum over its triangles of vOs projected into
//      the corner normal's plane and normalised, then normalised;
//   7. the corner's tangent = that vector, w = +1 for an orientation-preserving group, else -1.
//
// Output order and values are what Geometry::set_tangent_encoded receives: [x, y, z, w] per (face, corner); a corner no group
// reached keeps the initial (1, 0, 0, -1).  The caller then averages per shared vertex as tangents.rs:170-211,295-312 does.
#pragma once
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace awsm_mikk {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline bool not_zero(float f) { return std::fabs(f) > FLT_MIN; }
inline bool v_not_zero(V3 v) { return not_zero(v.x) || not_zero(v.y) || not_zero(v.z); }
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 normalized(V3 v) { return (1.0f / length(v)) * v; }
inline V3 project_unit(V3 v, V3 n) { V3 r = v - dot(n, v) * n; if (v_not_zero(r)) r = normalized(r); return r; }
inline bool v_eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

enum : uint32_t { kDegenerate = 1u, kGroupWithAny = 4u, kOrientPreserving = 8u };

struct TriInfo {
    int neighbour[3] = {-1, -1, -1};
    int group[3] = {-1, -1, -1};
    V3 os = {0, 0, 0}, ot = {0, 0, 0};
    float mag_s = 0.0f, mag_t = 0.0f;
    uint32_t flags = kGroupWithAny;
};
struct Group { uint32_t vertex; bool orient; std::vector<int> faces; };

// positions / normals: 3 floats per vertex; uvs: 2 floats per vertex; tris: 3 vertex indices per triangle (all < n_vertices).
// out: 4 floats per (triangle, corner).
inline void generate(const float* positions, const float* normals, const float* uvs, const uint32_t* tris, size_t n_tris, std::vector<std::array<float, 4>>& out) {
    out.assign(n_tris * 3, std::array<float, 4>{1.0f, 0.0f, 0.0f, -1.0f});     // STSpace initial value: vOs = (1, 0, 0), bOrient = false
    if (!n_tris) return;
    auto P = [&](uint32_t v) { return V3{positions[v * 3], positions[v * 3 + 1], positions[v * 3 + 2]}; };
    auto N = [&](uint32_t v) { return V3{normals[v * 3], normals[v * 3 + 1], normals[v * 3 + 2]}; };

    // 1. weld: the smallest corner's vertex stands for every corner with the same (position, normal, uv) values
    std::vector<uint32_t> id(n_tris * 3);
    {
        std::map<std::array<uint32_t, 8>, uint32_t> seen;
        for (size_t c = 0; c < n_tris * 3; c++) {
            const uint32_t v = tris[c];
            float k[8] = {positions[v * 3], positions[v * 3 + 1], positions[v * 3 + 2], normals[v * 3], normals[v * 3 + 1], normals[v * 3 + 2], uvs[v * 2], uvs[v * 2 + 1]};
            std::array<uint32_t, 8> key;
            for (int i = 0; i < 8; i++) { if (k[i] == 0.0f) k[i] = 0.0f; std::memcpy(&key[i], &k[i], 4); }      // -0 == +0, as the float compare has it
            id[c] = seen.emplace(key, v).first->second;
        }
    }
    // 2. degenerate triangles
    std::vector<TriInfo> ti(n_tris);
    for (size_t t = 0; t < n_tris; t++) {
        const V3 p0 = P(id[t * 3]), p1 = P(id[t * 3 + 1]), p2 = P(id[t * 3 + 2]);
        if (v_eq(p0, p1) || v_eq(p0, p2) || v_eq(p1, p2)) ti[t].flags |= kDegenerate;
    }
    auto good = [&](size_t t) { return !(ti[t].flags & kDegenerate); };
    // 3. per-triangle derivatives (InitTriInfo)
    for (size_t t = 0; t < n_tris; t++) {
        if (!good(t)) continue;
        const uint32_t i0 = id[t * 3], i1 = id[t * 3 + 1], i2 = id[t * 3 + 2];
        const V3 d1 = P(i1) - P(i0), d2 = P(i2) - P(i0);
        const float t21x = uvs[i1 * 2] - uvs[i0 * 2], t21y = uvs[i1 * 2 + 1] - uvs[i0 * 2 + 1], t31x = uvs[i2 * 2] - uvs[i0 * 2], t31y = uvs[i2 * 2 + 1] - uvs[i0 * 2 + 1];
        const float area2 = t21x * t31y - t21y * t31x;
        V3 os = (t31y * d1) - (t21y * d2), ot = (-t31x * d1) + (t21x * d2);
        if (area2 > 0.0f) ti[t].flags |= kOrientPreserving;
        if (not_zero(area2)) {
            const float abs_area = std::fabs(area2), len_os = length(os), len_ot = length(ot), s = (ti[t].flags & kOrientPreserving) ? 1.0f : -1.0f;
            if (not_zero(len_os)) os = (s / len_os) * os;
            if (not_zero(len_ot)) ot = (s / len_ot) * ot;
            ti[t].mag_s = len_os / abs_area; ti[t].mag_t = len_ot / abs_area;
            if (not_zero(ti[t].mag_s) && not_zero(ti[t].mag_t)) ti[t].flags &= ~kGroupWithAny;
        }
        ti[t].os = os; ti[t].ot = ot;
    }
    // 4. neighbours (BuildNeighborsFast): edges keyed by the sorted id pair, candidates in triangle order; a pair matches when the
    // two triangles run through the edge in opposite directions and neither side has a neighbour there yet
    {
        struct Edge { uint32_t a, b; uint32_t tri; };
        std::vector<Edge> edges;
        edges.reserve(n_tris * 3);
        for (size_t t = 0; t < n_tris; t++) {
            if (!good(t)) continue;
            for (int e = 0; e < 3; e++) { const uint32_t a = id[t * 3 + e], b = id[t * 3 + (e + 1) % 3]; edges.push_back({std::min(a, b), std::max(a, b), (uint32_t)t}); }
        }
        std::sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) { return x.a != y.a ? x.a < y.a : x.b != y.b ? x.b < y.b : x.tri < y.tri; });
        auto get_edge = [&](uint32_t t, uint32_t a, uint32_t b, uint32_t& o0, uint32_t& o1) {      // GetEdge: the edge of t with ids {a, b}, in t's winding
            const uint32_t* ix = &id[(size_t)t * 3];
            if (ix[0] == a || ix[0] == b) {
                if (ix[1] == a || ix[1] == b) { o0 = ix[0]; o1 = ix[1]; return 0; }
                o0 = ix[2]; o1 = ix[0]; return 2;
            }
            o0 = ix[1]; o1 = ix[2]; return 1;
        };
        for (size_t i = 0; i < edges.size(); i++) {
            uint32_t a0, a1;
            const int ea = get_edge(edges[i].tri, edges[i].a, edges[i].b, a0, a1);
            if (ti[edges[i].tri].neighbour[ea] != -1) continue;
            for (size_t j = i + 1; j < edges.size() && edges[j].a == edges[i].a && edges[j].b == edges[i].b; j++) {
                uint32_t b1, b0;
                const int eb = get_edge(edges[j].tri, edges[j].a, edges[j].b, b1, b0);      // flipped
                if (a0 == b0 && a1 == b1 && ti[edges[j].tri].neighbour[eb] == -1) {
                    ti[edges[i].tri].neighbour[ea] = (int)edges[j].tri;
                    ti[edges[j].tri].neighbour[eb] = (int)edges[i].tri;
                    break;
                }
            }
        }
    }
    // 5. groups (Build4RuleGroups / AssignRecur, the recursion unrolled onto a stack in the same visiting order: left edge first)
    std::vector<Group> groups;
    for (size_t f = 0; f < n_tris; f++) {
        if (!good(f)) continue;
        for (int i = 0; i < 3; i++) {
            if ((ti[f].flags & kGroupWithAny) || ti[f].group[i] != -1) continue;
            const int g = (int)groups.size();
            groups.push_back({id[f * 3 + i], (ti[f].flags & kOrientPreserving) != 0, {(int)f}});
            ti[f].group[i] = g;
            std::vector<int> stack;
            auto push_neighbours = [&](size_t t, int corner) {      // pushed right then left, so that the left one is visited first
                const int l = ti[t].neighbour[corner], r = ti[t].neighbour[corner > 0 ? corner - 1 : 2];
                if (r >= 0) stack.push_back(r);
                if (l >= 0) stack.push_back(l);
            };
            push_neighbours(f, i);
            while (!stack.empty()) {
                const int t = stack.back(); stack.pop_back();
                int c = -1;
                for (int k = 0; k < 3; k++) if (id[(size_t)t * 3 + k] == groups[g].vertex) { c = k; break; }
                if (c < 0 || ti[t].group[c] != -1) continue;                            // already in this group (or in another one)
                if (ti[t].flags & kGroupWithAny) {                                      // the first group to reach such a triangle decides its orientation
                    if (ti[t].group[0] == -1 && ti[t].group[1] == -1 && ti[t].group[2] == -1)
                        ti[t].flags = (ti[t].flags & ~kOrientPreserving) | (groups[g].orient ? kOrientPreserving : 0u);
                }
                if (((ti[t].flags & kOrientPreserving) != 0) != groups[g].orient) continue;
                groups[g].faces.push_back(t);
                ti[t].group[c] = g;
                push_neighbours((size_t)t, c);
            }
        }
    }
    // 6. + 7. tangent spaces (GenerateTSpaces / EvalTspace), threshold 180 degrees
    const float thres_cos = std::cos(180.0f * 3.14159265358979323846f / 180.0f);
    auto eval = [&](const std::vector<int>& members, uint32_t vertex) {
        V3 os = {0, 0, 0};
        for (int t : members) {
            if (ti[t].flags & kGroupWithAny) continue;                                  // only triangles with a valid UV map contribute
            int i = -1;
            for (int k = 0; k < 3; k++) if (id[(size_t)t * 3 + k] == vertex) i = k;
            if (i < 0) continue;
            const V3 n = N(id[(size_t)t * 3 + i]);
            const V3 v_os = project_unit(ti[t].os, n);
            const uint32_t i2 = id[(size_t)t * 3 + (i < 2 ? i + 1 : 0)], i1 = id[(size_t)t * 3 + i], i0 = id[(size_t)t * 3 + (i > 0 ? i - 1 : 2)];
            const V3 v1 = project_unit(P(i0) - P(i1), n), v2 = project_unit(P(i2) - P(i1), n);
            float c = dot(v1, v2);
            c = c > 1.0f ? 1.0f : (c < -1.0f ? -1.0f : c);
            const float angle = (float)std::acos((double)c);
            os = os + angle * v_os;
        }
        if (v_not_zero(os)) os = normalized(os);
        return os;
    };
    for (const Group& g : groups) {
        std::vector<std::pair<std::vector<int>, V3>> subs;                               // unique sub-groups of this group and their result
        for (int f : g.faces) {
            int index = -1;
            for (int k = 0; k < 3; k++) if (ti[f].group[k] == (int)(&g - groups.data())) index = k;
            if (index < 0) continue;
            const V3 n = N(id[(size_t)f * 3 + index]);
            const V3 os = project_unit(ti[f].os, n), ot = project_unit(ti[f].ot, n);
            std::vector<int> members;
            for (int t : g.faces) {
                const V3 os2 = project_unit(ti[t].os, n), ot2 = project_unit(ti[t].ot, n);
                const bool any = ((ti[f].flags | ti[t].flags) & kGroupWithAny) != 0;
                if (any || t == f || (dot(os, os2) > thres_cos && dot(ot, ot2) > thres_cos)) members.push_back(t);
            }
            std::sort(members.begin(), members.end());
            size_t l = 0;
            while (l < subs.size() && subs[l].first != members) l++;
            if (l == subs.size()) subs.emplace_back(members, eval(members, g.vertex));
            const V3 r = subs[l].second;
            out[(size_t)f * 3 + index] = {r.x, r.y, r.z, g.orient ? 1.0f : -1.0f};
        }
    }
    // degenerate triangles (DegenEpilogue): every corner copies the first corner of a good triangle with the same welded id
    bool any_degenerate = false;
    for (size_t t = 0; t < n_tris; t++) any_degenerate |= !good(t);
    if (any_degenerate) {
        std::map<uint32_t, size_t> first;                                               // welded id -> first good corner
        for (size_t t = 0; t < n_tris; t++) if (good(t)) for (int k = 0; k < 3; k++) first.emplace(id[t * 3 + k], t * 3 + k);
        for (size_t t = 0; t < n_tris; t++) {
            if (good(t)) continue;
            for (int k = 0; k < 3; k++) { auto it = first.find(id[t * 3 + k]); if (it != first.end()) out[t * 3 + k] = out[it->second]; }
        }
    }
}

}  // namespace awsm_mikk
