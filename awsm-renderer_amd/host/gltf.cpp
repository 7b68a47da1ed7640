// gltf.cpp — native glTF 2.0 / GLB reader for the host layer: awsm_host_load_gltf (include/awsm_host.h).
//
// Replaces, for the inputs of the hot path, crates/renderer/src/gltf/{loader,data,buffers,populate}.rs and the `gltf` crate
// underneath them.  It produces exactly what the rest of the host layer consumes — decoded RGBA8 images, samplers, materials,
// the node hierarchy, skins, and per primitive positions / normals / tangents / UV + colour sets / joints + weights / morph
// targets / triangle indices — and feeds them through the same key API in the reference's populate order
// (populate.rs:185-205: every transform, then skins, then meshes), so a scene loaded from a file and the same scene described
// through the API leave byte-identical mirrors.
//
// Per primitive, as gltf/buffers/mesh.rs does: indices are read (u8/u16/u32) or generated, triangle strips and fans become
// lists (buffers/index.rs:116-205); missing normals are accumulated from face normals (buffers/normals.rs:46-126); missing
// tangents are generated when the material has a normal map (buffers/tangents.rs:11-98): the reference calls
// bevy_mikktspace there and then averages its per-corner output per shared vertex (tangents.rs:165-205,295-312); mikktspace.hpp
// restates the crate's algorithm (welding, orientation groups around each vertex, angle-weighted evaluation), compute_tangents
// below the averaging, the fallbacks and the sign vote.
//
// Images: PNG (png.hpp, zlib inflate) and JPEG (jpeg.hpp: baseline, extended sequential and progressive; Huffman, 8 bit).  KTX2 and anything else return
// AWSM_ERR_UNSUPPORTED with the image index in the message.
// Not read: cameras (the caller owns the camera), animations, KHR_mesh_quantization beyond the normalised integer attribute
// types glTF core already allows.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/awsm_host.h"
#include "json.hpp"
#include "mikktspace.hpp"
#include "jpeg.hpp"
#include "png.hpp"

using awsm_json::Value;

namespace {

struct Loader {
    AwsmHost* h = nullptr;
    std::string err;
    std::string dir;                                  // directory of the .gltf, for relative URIs
    Value doc;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<int> image_tex;                       // glTF image -> host texture id
    std::map<std::vector<uint32_t>, uint32_t> sampler_ids;   // AwsmSampler fields -> host sampler id
    std::vector<AwsmKey> node_keys;
    std::map<int, AwsmKey> material_keys;             // glTF material (-1 = default) -> key
    std::map<std::vector<float>, AwsmKey> tex_transform_keys;
    AwsmGltfInfo info{};

    bool fail(const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return false;
    }
};

bool err_or(Loader& L, const char* msg);

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    const size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

bool base64_decode(const char* s, size_t n, std::vector<uint8_t>& out) {
    out.clear();
    uint32_t acc = 0; int bits = 0;
    for (size_t i = 0; i < n; i++) {
        const char c = s[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=' || c == '\n' || c == '\r') continue;
        else return false;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xFF)); }
    }
    return true;
}

// data: URI or a path relative to the document
bool load_uri(Loader& L, const std::string& uri, std::vector<uint8_t>& out) {
    if (uri.compare(0, 5, "data:") == 0) {
        const size_t comma = uri.find(',');
        if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma) return L.fail("data URI without base64 payload");
        if (!base64_decode(uri.c_str() + comma + 1, uri.size() - comma - 1, out)) return L.fail("bad base64 in data URI");
        return true;
    }
    std::string path;   // percent-decoding of the few characters exporters escape
    for (size_t i = 0; i < uri.size(); i++) {
        if (uri[i] == '%' && i + 2 < uri.size()) { path += (char)strtol(uri.substr(i + 1, 2).c_str(), nullptr, 16); i += 2; }
        else path += uri[i];
    }
    if (!read_file(L.dir + path, out)) return L.fail("cannot read '%s'", (L.dir + path).c_str());
    return true;
}

// ---- accessors (buffers/accessor.rs): any component type / normalisation / stride -> tightly packed f32 or u32 ----
int type_components(const std::string& t) {
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4" || t == "MAT2") return 4;
    if (t == "MAT3") return 9;
    if (t == "MAT4") return 16;
    return 0;
}
int component_size(int ct) { return ct == 5120 || ct == 5121 ? 1 : (ct == 5122 || ct == 5123 ? 2 : (ct == 5125 || ct == 5126 ? 4 : 0)); }

struct AccessorView {
    const uint8_t* base = nullptr; size_t stride = 0, count = 0; int comps = 0, ctype = 0; bool normalized = false;
    std::vector<uint8_t> owned;     // sparse / zero-filled accessors: the tightly packed element bytes (accessor.rs:14-66)
};

// A JSON number that must be a byte offset / size / count: non-negative, integral, below 2^53 (json.hpp keeps doubles; integer() of a
// negative or huge value used to be cast to size_t and wrap).
bool nonneg(Loader& L, const Value& v, const char* what, int index, size_t dflt, size_t* out) {
    if (!v.is_number()) { *out = dflt; return true; }
    const double d = v.number(0.0);
    if (!(d >= 0.0) || d > 9007199254740992.0 || d != std::floor(d)) return L.fail("%s %d: offset / size / count %g is not a non-negative integer", what, index, d);
    *out = (size_t)d;
    return true;
}
// [off, off + len) inside a buffer of `size` bytes, without wrapping
bool fits(size_t off, size_t len, size_t size) { return off <= size && len <= size - off; }

// bufferView `bvi` -> buffer bytes, start offset and length, all checked against the buffer
bool buffer_view(Loader& L, const Value& bvv, const char* what, int index, const uint8_t** data, size_t* len, size_t* stride) {
    if (!bvv.is_number()) return L.fail("%s %d: bufferView missing", what, index);
    const double bvd = bvv.number(-1.0);
    const Value& views = L.doc["bufferViews"];
    if (!(bvd >= 0.0) || bvd >= (double)views.size() || bvd != std::floor(bvd)) return L.fail("%s %d: bufferView index out of range", what, index);
    const Value& bv = views[(size_t)bvd];
    if (!bv.is_object()) return L.fail("%s %d: bufferView missing", what, index);
    const double bid = bv["buffer"].number(-1.0);
    if (!(bid >= 0.0) || bid >= (double)L.buffers.size() || bid != std::floor(bid)) return L.fail("%s %d: buffer missing", what, index);
    const std::vector<uint8_t>& buf = L.buffers[(size_t)bid];
    size_t off = 0, blen = 0, st = 0;
    if (!nonneg(L, bv["byteOffset"], what, index, 0, &off) || !nonneg(L, bv["byteStride"], what, index, 0, &st)) return false;
    if (!fits(off, 0, buf.size())) return L.fail("%s %d: bufferView starts beyond its buffer", what, index);
    if (!nonneg(L, bv["byteLength"], what, index, buf.size() - off, &blen)) return false;
    if (!fits(off, blen, buf.size())) return L.fail("%s %d: bufferView exceeds its buffer", what, index);
    if (st > 255) return L.fail("%s %d: byteStride %zu above the 252 the format allows", what, index, st);
    *data = buf.data() + off; *len = blen;
    if (stride) *stride = st;
    return true;
}

bool accessor_view(Loader& L, int index, AccessorView& v) {
    const Value& a = L.doc["accessors"][(size_t)index];
    if (index < 0 || !a.is_object()) return L.fail("accessor %d missing", index);
    v.comps = type_components(a["type"].string());
    v.ctype = (int)a["componentType"].integer(0);
    v.normalized = a["normalized"].boolean(false);
    const int cs = component_size(v.ctype);
    if (!v.comps || !cs) return L.fail("accessor %d: bad type", index);
    if ((a["type"].string() == "MAT2" && cs == 1) || (a["type"].string() == "MAT3" && cs <= 2)) return L.fail("accessor %d: padded matrix layouts are not supported", index);
    if (!nonneg(L, a["count"], "accessor", index, 0, &v.count)) return false;
    const size_t elem = (size_t)v.comps * cs;
    if (v.count > (size_t)1 << 32) return L.fail("accessor %d: %zu elements", index, v.count);
    const bool sparse = a.has("sparse");
    if (a.has("bufferView")) {
        const uint8_t* data; size_t len, stride, aoff;
        if (!buffer_view(L, a["bufferView"], "accessor", index, &data, &len, &stride)) return false;
        if (!nonneg(L, a["byteOffset"], "accessor", index, 0, &aoff)) return false;
        v.stride = stride ? stride : elem;
        if (v.stride < elem) return L.fail("accessor %d: byteStride %zu below the element size %zu", index, v.stride, elem);
        // the last element ends at aoff + (count - 1) * stride + elem: inside the VIEW (which is inside its buffer)
        if (v.count && (!fits(aoff, elem, len) || (v.count - 1) > (len - aoff - elem) / v.stride)) return L.fail("accessor %d exceeds its bufferView", index);
        v.base = data + aoff;
    } else {
        // "if we have no view, fill it with zeroes" — which a sparse block or an extension may then overwrite (accessor.rs:37-43)
        v.owned.assign(v.count * elem, 0);
        v.base = v.owned.data(); v.stride = elem;
    }
    if (sparse) {   // accessor.rs:45-63: substitute `count` elements at the listed indices
        const Value& sp = a["sparse"];
        size_t n = 0;
        if (!nonneg(L, sp["count"], "sparse accessor", index, 0, &n)) return false;
        if (n > v.count) return L.fail("accessor %d: %zu sparse elements in an accessor of %zu", index, n, v.count);
        if (v.owned.empty() && v.count) {     // repack the dense data first (the view is borrowed, and may be strided)
            v.owned.resize(v.count * elem);
            for (size_t i = 0; i < v.count; i++) memcpy(v.owned.data() + i * elem, v.base + i * v.stride, elem);
        }
        v.base = v.owned.data(); v.stride = elem;
        const uint8_t *idata, *vdata; size_t ilen, vlen, ioff, voff;
        if (!buffer_view(L, sp["indices"]["bufferView"], "sparse indices of accessor", index, &idata, &ilen, nullptr)) return false;
        if (!buffer_view(L, sp["values"]["bufferView"], "sparse values of accessor", index, &vdata, &vlen, nullptr)) return false;
        if (!nonneg(L, sp["indices"]["byteOffset"], "sparse accessor", index, 0, &ioff) || !nonneg(L, sp["values"]["byteOffset"], "sparse accessor", index, 0, &voff)) return false;
        const int ict = (int)sp["indices"]["componentType"].integer(0);
        const size_t isz = ict == 5121 ? 1 : (ict == 5123 ? 2 : (ict == 5125 ? 4 : 0));
        if (!isz) return L.fail("accessor %d: sparse index type %d (u8 / u16 / u32 only)", index, ict);
        if (n && (!fits(ioff, n * isz, ilen) || !fits(voff, n * elem, vlen))) return L.fail("accessor %d: sparse data exceeds its bufferView", index);
        for (size_t k = 0; k < n; k++) {
            size_t target = 0;
            const uint8_t* p = idata + ioff + k * isz;
            if (isz == 1) target = p[0]; else if (isz == 2) { uint16_t t; memcpy(&t, p, 2); target = t; } else { uint32_t t; memcpy(&t, p, 4); target = t; }
            if (target >= v.count) return L.fail("accessor %d: sparse index %zu outside its %zu elements", index, target, v.count);
            memcpy(v.owned.data() + target * elem, vdata + voff + k * elem, elem);
        }
    }
    return true;
}

float component_f32(const uint8_t* p, int ctype, bool normalized) {
    switch (ctype) {
        case 5126: { float f; memcpy(&f, p, 4); return f; }
        case 5121: return normalized ? (float)p[0] / 255.0f : (float)p[0];
        case 5120: { const int8_t v = (int8_t)p[0]; return normalized ? std::max((float)v / 127.0f, -1.0f) : (float)v; }
        case 5123: { uint16_t v; memcpy(&v, p, 2); return normalized ? (float)v / 65535.0f : (float)v; }
        case 5122: { int16_t v; memcpy(&v, p, 2); return normalized ? std::max((float)v / 32767.0f, -1.0f) : (float)v; }
        case 5125: { uint32_t v; memcpy(&v, p, 4); return (float)v; }
    }
    return 0.0f;
}

bool read_floats(Loader& L, int accessor, int want_comps, std::vector<float>& out, size_t* count, bool raw_integers = false) {
    AccessorView v;
    if (!accessor_view(L, accessor, v)) return false;
    if (raw_integers) v.normalized = false;      // `value as f32`, whatever the accessor's normalized flag says
    if (v.comps != want_comps) return L.fail("accessor %d: expected %d components, found %d", accessor, want_comps, v.comps);
    const int cs = component_size(v.ctype);
    out.resize(v.count * (size_t)want_comps);
    for (size_t i = 0; i < v.count; i++)
        for (int c = 0; c < want_comps; c++) out[i * want_comps + c] = component_f32(v.base + i * v.stride + (size_t)c * cs, v.ctype, v.normalized);
    if (count) *count = v.count;
    return true;
}

bool read_uints(Loader& L, int accessor, int want_comps, std::vector<uint32_t>& out, size_t* count) {
    AccessorView v;
    if (!accessor_view(L, accessor, v)) return false;
    if (v.comps != want_comps) return L.fail("accessor %d: expected %d components, found %d", accessor, want_comps, v.comps);
    if (v.ctype == 5126) return L.fail("accessor %d: float where integers are required", accessor);
    const int cs = component_size(v.ctype);
    out.resize(v.count * (size_t)want_comps);
    for (size_t i = 0; i < v.count; i++)
        for (int c = 0; c < want_comps; c++) {
            const uint8_t* p = v.base + i * v.stride + (size_t)c * cs;
            uint32_t x = 0;
            if (cs == 1) x = p[0]; else if (cs == 2) { uint16_t t; memcpy(&t, p, 2); x = t; } else memcpy(&x, p, 4);
            out[i * want_comps + c] = x;
        }
    if (count) *count = v.count;
    return true;
}

// ---- images and samplers ----
bool image_bytes(Loader& L, const Value& img, std::vector<uint8_t>& out) {
    if (img.has("bufferView")) {
        const uint8_t* data; size_t len;
        if (!buffer_view(L, img["bufferView"], "image", 0, &data, &len, nullptr)) return false;
        out.assign(data, data + len);
        return true;
    }
    if (img.has("uri")) return load_uri(L, img["uri"].string(), out);
    return L.fail("image without uri or bufferView");
}

// MipmapTextureKind by the role under which an image first enters the pool: materials in order, roles in the order
// pbr_material_mapper visits them (gltf/populate/material.rs:94-640).  0 albedo, 1 normal, 2 metallic-roughness,
// 3 occlusion, 4 emissive, 5 specular, 7 transmission, 8 volume thickness.
void assign_image_kinds(Loader& L, std::vector<int>& kinds) {
    const Value& textures = L.doc["textures"];
    auto use = [&](const Value& texinfo, int kind) {
        if (!texinfo.is_object()) return;
        const int64_t ti = texinfo["index"].integer(-1);
        if (ti < 0 || (size_t)ti >= textures.size()) return;
        const int64_t src = textures[(size_t)ti]["source"].integer(-1);
        if (src >= 0 && (size_t)src < kinds.size() && kinds[(size_t)src] < 0) kinds[(size_t)src] = kind;
    };
    const Value& mats = L.doc["materials"];
    for (size_t i = 0; i < mats.size(); i++) {
        const Value& m = mats[i];
        const Value& pbr = m["pbrMetallicRoughness"];
        const Value& ext = m["extensions"];
        use(pbr["baseColorTexture"], 0);
        use(pbr["metallicRoughnessTexture"], 2);
        use(m["normalTexture"], 1);
        use(m["occlusionTexture"], 3);
        use(m["emissiveTexture"], 4);
        use(ext["KHR_materials_specular"]["specularTexture"], 5);
        use(ext["KHR_materials_specular"]["specularColorTexture"], 5);
        use(ext["KHR_materials_transmission"]["transmissionTexture"], 7);
        use(ext["KHR_materials_volume"]["thicknessTexture"], 8);
        use(ext["KHR_materials_clearcoat"]["clearcoatTexture"], 0);
        use(ext["KHR_materials_clearcoat"]["clearcoatRoughnessTexture"], 2);
        use(ext["KHR_materials_clearcoat"]["clearcoatNormalTexture"], 1);
        use(ext["KHR_materials_sheen"]["sheenColorTexture"], 5);
        use(ext["KHR_materials_sheen"]["sheenRoughnessTexture"], 2);
    }
}

bool load_images(Loader& L) {
    const Value& images = L.doc["images"];
    std::vector<int> kinds(images.size(), -1);
    assign_image_kinds(L, kinds);
    L.image_tex.assign(images.size(), -1);
    for (size_t i = 0; i < images.size(); i++) {
        std::vector<uint8_t> bytes, rgba;
        if (!image_bytes(L, images[i], bytes)) return false;
        uint32_t w = 0, ht = 0;
        std::string perr;
        if (awsm_png::is_png(bytes.data(), bytes.size())) { if (!awsm_png::decode(bytes.data(), bytes.size(), rgba, w, ht, perr)) return L.fail("image %zu: %s", i, perr.c_str()); }
        else if (awsm_jpeg::is_jpeg(bytes.data(), bytes.size())) { if (!awsm_jpeg::decode(bytes.data(), bytes.size(), rgba, w, ht, perr)) return L.fail("image %zu: %s", i, perr.c_str()); }
        else return L.fail("image %zu: this image format is not supported (PNG and baseline JPEG only)", i);
        const int id = awsm_host_texture_insert_kind(L.h, rgba.data(), w, ht, (uint32_t)(kinds[i] < 0 ? 0 : kinds[i]));
        if (id < 0) return L.fail("image %zu: %s", i, awsm_host_last_error(L.h));
        L.image_tex[i] = id;
        L.info.images++;
    }
    return true;
}

// gltf/populate/material.rs:886-981: linear / linear / linear + 16x anisotropy unless the glTF sampler says otherwise; repeat
// unless it says otherwise; anisotropy only when all three filters are linear.  One host sampler per distinct result.
bool sampler_for_texture(Loader& L, const Value& tex, uint32_t* out) {
    AwsmSampler s{};
    s.address_mode_u = 1; s.address_mode_v = 1; s.mag_filter = 1; s.min_filter = 1; s.mipmap_filter = 1; s.max_anisotropy = 16;
    const int64_t si = tex["sampler"].integer(-1);
    if (si >= 0) {
        const Value& g = L.doc["samplers"][(size_t)si];
        auto wrap = [](int64_t v) -> uint32_t { return v == 33071 ? 0u : (v == 33648 ? 2u : 1u); };
        s.address_mode_u = wrap(g["wrapS"].integer(10497));
        s.address_mode_v = wrap(g["wrapT"].integer(10497));
        const int64_t mag = g["magFilter"].integer(-1), mn = g["minFilter"].integer(-1);
        if (mag == 9728) s.mag_filter = 0; else if (mag == 9729) s.mag_filter = 1;
        switch (mn) {
            case 9728: s.min_filter = 0; break;
            case 9729: s.min_filter = 1; break;
            case 9984: s.min_filter = 0; s.mipmap_filter = 0; break;
            case 9985: s.min_filter = 1; s.mipmap_filter = 0; break;
            case 9986: s.min_filter = 0; s.mipmap_filter = 1; break;
            case 9987: s.min_filter = 1; s.mipmap_filter = 1; break;
            default: break;
        }
        if (g["extras"].has("max_anisotropy")) s.max_anisotropy = (uint32_t)g["extras"]["max_anisotropy"].integer(16);   // round trip of scenes exported by this repo
    }
    if (!(s.mag_filter == 1 && s.min_filter == 1 && s.mipmap_filter == 1)) s.max_anisotropy = 1;   // SamplerCacheKey::allowed_ansiotropy
    const std::vector<uint32_t> key = {s.address_mode_u, s.address_mode_v, s.mag_filter, s.min_filter, s.mipmap_filter, s.max_anisotropy};
    auto it = L.sampler_ids.find(key);
    if (it == L.sampler_ids.end()) {
        const int id = awsm_host_sampler_insert(L.h, &s);
        if (id < 0) return L.fail("sampler: %s", awsm_host_last_error(L.h));
        it = L.sampler_ids.emplace(key, (uint32_t)id).first;
        L.info.samplers++;
    }
    *out = it->second;
    return true;
}

// samplers are created in document order of the textures, so that ids do not depend on which material is visited first
bool load_samplers(Loader& L) {
    const Value& textures = L.doc["textures"];
    for (size_t i = 0; i < textures.size(); i++) { uint32_t id; if (!sampler_for_texture(L, textures[i], &id)) return false; }
    return true;
}

bool tex_ref(Loader& L, const Value& info, AwsmHostTexRef& r) {
    r.texture = -1; r.sampler = 0; r.uv_index = 0; r.pad = 0; r.transform = 0;
    if (!info.is_object()) return true;
    const int64_t ti = info["index"].integer(-1);
    const Value& tex = L.doc["textures"][(size_t)ti];
    if (ti < 0 || !tex.is_object()) return true;                         // dangling reference: SkipTexture (materials/writer.rs:100-112)
    const int64_t src = tex["source"].integer(-1);
    if (src < 0 || (size_t)src >= L.image_tex.size()) return true;
    r.texture = L.image_tex[(size_t)src];
    if (!sampler_for_texture(L, tex, &r.sampler)) return false;
    r.uv_index = (uint32_t)info["texCoord"].integer(0);
    const Value& xf = info["extensions"]["KHR_texture_transform"];
    if (xf.is_object()) {
        float offset[2] = {(float)xf["offset"][(size_t)0].number(0.0), (float)xf["offset"][(size_t)1].number(0.0)};
        float scale[2] = {(float)xf["scale"][(size_t)0].number(1.0), (float)xf["scale"][(size_t)1].number(1.0)};
        float origin[2] = {(float)xf["extras"]["origin"][(size_t)0].number(0.0), (float)xf["extras"]["origin"][(size_t)1].number(0.0)};   // the reference's TextureTransform has an origin; glTF does not
        const float rotation = (float)xf["rotation"].number(0.0);
        if (xf.has("texCoord")) r.uv_index = (uint32_t)xf["texCoord"].integer(r.uv_index);
        const std::vector<float> key = {offset[0], offset[1], origin[0], origin[1], rotation, scale[0], scale[1]};
        auto it = L.tex_transform_keys.find(key);
        if (it == L.tex_transform_keys.end()) it = L.tex_transform_keys.emplace(key, awsm_host_texture_transform_insert(L.h, offset, origin, rotation, scale)).first;
        r.transform = it->second;
    }
    return true;
}

void vec_n(const Value& v, float* out, int n, const float* dflt) { for (int i = 0; i < n; i++) out[i] = (float)v[(size_t)i].number(dflt[i]); }

// gltf/populate/material.rs:94-640
bool material_key(Loader& L, int index, AwsmKey* out) {
    auto it = L.material_keys.find(index);
    if (it != L.material_keys.end()) { *out = it->second; return true; }
    const Value& m = index >= 0 ? L.doc["materials"][(size_t)index] : Value();
    const Value& pbr = m["pbrMetallicRoughness"];
    const Value& ext = m["extensions"];
    AwsmHostMaterial hm;
    memset(&hm, 0, sizeof hm);
    hm.struct_size = (uint32_t)sizeof hm;
    AwsmHostTexRef none; none.texture = -1; none.sampler = 0; none.uv_index = 0; none.pad = 0; none.transform = 0;
    hm.base_color_tex = hm.metallic_roughness_tex = hm.normal_tex = hm.occlusion_tex = hm.emissive_tex = none;
    hm.specular_tex = hm.specular_color_tex = hm.transmission_tex = hm.volume_thickness_tex = none;
    hm.clearcoat_tex = hm.clearcoat_roughness_tex = hm.clearcoat_normal_tex = hm.sheen_roughness_tex = hm.sheen_color_tex = none;
    hm.shader = ext.has("KHR_materials_unlit") ? 2u : 1u;
    hm.double_sided = m["doubleSided"].boolean(false) ? 1u : 0u;
    const float one4[4] = {1, 1, 1, 1}, zero3[3] = {0, 0, 0}, one3[3] = {1, 1, 1};
    vec_n(pbr["baseColorFactor"], hm.base_color_factor, 4, one4);
    hm.metallic_factor = (float)pbr["metallicFactor"].number(1.0);
    hm.roughness_factor = (float)pbr["roughnessFactor"].number(1.0);
    hm.normal_scale = (float)m["normalTexture"]["scale"].number(m["extras"]["normal_scale"].number(1.0));           // extras: a factor kept without its texture
    hm.occlusion_strength = (float)m["occlusionTexture"]["strength"].number(m["extras"]["occlusion_strength"].number(1.0));
    vec_n(m["emissiveFactor"], hm.emissive_factor, 3, zero3);
    hm.debug_bitmask = (uint32_t)m["extras"]["debug_bitmask"].integer(0);
    const std::string& am = m["alphaMode"].string();
    hm.alpha_mode = am == "MASK" ? 1u : (am == "BLEND" ? 2u : 0u);
    hm.alpha_cutoff = (float)m["alphaCutoff"].number(0.5);
    if (!tex_ref(L, pbr["baseColorTexture"], hm.base_color_tex) || !tex_ref(L, pbr["metallicRoughnessTexture"], hm.metallic_roughness_tex) ||
        !tex_ref(L, m["normalTexture"], hm.normal_tex) || !tex_ref(L, m["occlusionTexture"], hm.occlusion_tex) || !tex_ref(L, m["emissiveTexture"], hm.emissive_tex)) return false;
    if (m["extras"].has("vertex_color_set")) { hm.has_vertex_color = 1; hm.vertex_color_set = (uint32_t)m["extras"]["vertex_color_set"].integer(0); }
    if (ext.has("KHR_materials_emissive_strength")) { hm.has_emissive_strength = 1; hm.emissive_strength = (float)ext["KHR_materials_emissive_strength"]["emissiveStrength"].number(1.0); }
    if (ext.has("KHR_materials_ior")) { hm.has_ior = 1; hm.ior = (float)ext["KHR_materials_ior"]["ior"].number(1.5); }
    if (ext.has("KHR_materials_specular")) {
        const Value& e = ext["KHR_materials_specular"];
        hm.has_specular = 1; hm.specular_factor = (float)e["specularFactor"].number(1.0);
        vec_n(e["specularColorFactor"], hm.specular_color_factor, 3, one3);
        if (!tex_ref(L, e["specularTexture"], hm.specular_tex) || !tex_ref(L, e["specularColorTexture"], hm.specular_color_tex)) return false;
    }
    if (ext.has("KHR_materials_transmission")) {
        const Value& e = ext["KHR_materials_transmission"];
        hm.has_transmission = 1; hm.transmission_factor = (float)e["transmissionFactor"].number(0.0);
        if (!tex_ref(L, e["transmissionTexture"], hm.transmission_tex)) return false;
    }
    if (ext.has("KHR_materials_volume")) {
        const Value& e = ext["KHR_materials_volume"];
        hm.has_volume = 1; hm.volume_thickness_factor = (float)e["thicknessFactor"].number(0.0);
        hm.volume_attenuation_distance = (float)e["attenuationDistance"].number(0.0);     // glTF's default is +inf; the reference stores 0 for "none"
        vec_n(e["attenuationColor"], hm.volume_attenuation_color, 3, one3);
        if (!tex_ref(L, e["thicknessTexture"], hm.volume_thickness_tex)) return false;
    }
    if (ext.has("KHR_materials_clearcoat")) {
        const Value& e = ext["KHR_materials_clearcoat"];
        hm.has_clearcoat = 1; hm.clearcoat_factor = (float)e["clearcoatFactor"].number(0.0);
        hm.clearcoat_roughness_factor = (float)e["clearcoatRoughnessFactor"].number(0.0);
        hm.clearcoat_normal_scale = (float)e["clearcoatNormalTexture"]["scale"].number(e["extras"]["normal_scale"].number(1.0));
        if (!tex_ref(L, e["clearcoatTexture"], hm.clearcoat_tex) || !tex_ref(L, e["clearcoatRoughnessTexture"], hm.clearcoat_roughness_tex) ||
            !tex_ref(L, e["clearcoatNormalTexture"], hm.clearcoat_normal_tex)) return false;
    }
    if (ext.has("KHR_materials_sheen")) {
        const Value& e = ext["KHR_materials_sheen"];
        hm.has_sheen = 1; hm.sheen_roughness_factor = (float)e["sheenRoughnessFactor"].number(0.0);
        vec_n(e["sheenColorFactor"], hm.sheen_color_factor, 3, zero3);
        if (!tex_ref(L, e["sheenRoughnessTexture"], hm.sheen_roughness_tex) || !tex_ref(L, e["sheenColorTexture"], hm.sheen_color_tex)) return false;
    }
    const AwsmKey k = awsm_host_material_insert(L.h, &hm);
    if (!k) return L.fail("material %d: %s", index, awsm_host_last_error(L.h));
    L.material_keys[index] = k;
    L.info.materials++;
    *out = k;
    return true;
}

// ---- geometry helpers ----
struct V3 { float x, y, z; };
V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
V3 normalize3(V3 v) { const float l2 = dot(v, v); if (l2 > 1e-20f) { const float inv = 1.0f / std::sqrt(l2); return scale(v, inv); } return {0, 0, 0}; }   // tangents.rs:223-231

// buffers/normals.rs:46-126: face normals (not normalised: area weighting) accumulated per vertex, then normalised; zero stays zero
void compute_normals(const std::vector<float>& pos, const std::vector<uint32_t>& idx, std::vector<float>& nrm) {
    const size_t V = pos.size() / 3;
    nrm.assign(V * 3, 0.0f);
    auto P = [&](uint32_t i) { return V3{pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]}; };
    for (size_t t = 0; t + 2 < idx.size(); t += 3) {
        const V3 fn = cross(sub(P(idx[t + 1]), P(idx[t])), sub(P(idx[t + 2]), P(idx[t])));
        for (int c = 0; c < 3; c++) { float* n = &nrm[(size_t)idx[t + c] * 3]; n[0] += fn.x; n[1] += fn.y; n[2] += fn.z; }
    }
    for (size_t v = 0; v < V; v++) {
        const V3 n = {nrm[v * 3], nrm[v * 3 + 1], nrm[v * 3 + 2]};
        if (n.x != 0.0f || n.y != 0.0f || n.z != 0.0f) { const float inv = 1.0f / std::sqrt(dot(n, n)); nrm[v * 3] = n.x * inv; nrm[v * 3 + 1] = n.y * inv; nrm[v * 3 + 2] = n.z * inv; }   // glam normalize
    }
}

// buffers/tangents.rs:268-364: mikktspace per triangle corner (mikktspace.hpp), then MikkTSpaceGeometry::set_tangent_encoded's
// accumulation per shared vertex (tangents.rs:295-312) and finalize_tangents (tangents.rs:170-211)
void compute_tangents(const std::vector<float>& pos, const std::vector<float>& nrm, const std::vector<float>& uv, const std::vector<uint32_t>& idx, std::vector<float>& tan) {
    const size_t V = pos.size() / 3, T = idx.size() / 3;
    std::vector<V3> sum(V, V3{0, 0, 0});
    std::vector<float> sign_sum(V, 0.0f);
    std::vector<uint32_t> pos_count(V, 0), neg_count(V, 0), count(V, 0);
    std::vector<std::array<float, 4>> corner;
    awsm_mikk::generate(pos.data(), nrm.data(), uv.data(), idx.data(), T, corner);
    for (size_t c = 0; c < T * 3; c++) {
        const uint32_t v = idx[c];
        sum[v].x += corner[c][0]; sum[v].y += corner[c][1]; sum[v].z += corner[c][2];
        sign_sum[v] += corner[c][3];
        if (corner[c][3] > 0.0f) pos_count[v]++; else if (corner[c][3] < 0.0f) neg_count[v]++;
        count[v]++;
    }
    tan.assign(V * 4, 0.0f);
    for (size_t v = 0; v < V; v++) {
        float* o = &tan[v * 4];
        if (count[v] == 0) { o[0] = 1.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 1.0f; continue; }     // tangents.rs:168-171
        const V3 n = normalize3({nrm[v * 3], nrm[v * 3 + 1], nrm[v * 3 + 2]});
        V3 t = normalize3(sub(sum[v], scale(n, dot(sum[v], n))));                                 // normalize_or_fallback
        if (!(dot(t, t) > 0.0f)) {                                                                // canonical_tangent_from_normal
            const V3 axis = std::fabs(n.y) < 0.999f ? V3{0, 1, 0} : V3{1, 0, 0};
            t = normalize3(cross(axis, n));
            if (!(dot(t, t) > 0.0f)) t = {1, 0, 0};
        }
        if (!(std::isfinite(t.x) && std::isfinite(t.y) && std::isfinite(t.z))) t = {1, 0, 0};
        float sgn;
        if (!std::isfinite(sign_sum[v])) sgn = 1.0f;
        else if (std::fabs(sign_sum[v]) >= 1e-4f) sgn = sign_sum[v] > 0.0f ? 1.0f : -1.0f;
        else sgn = pos_count[v] >= neg_count[v] ? 1.0f : -1.0f;
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = sgn;
    }
}

// ---- nodes ----
void mat4_decompose(const float m[16], float t[3], float r[4], float s[3]) {   // glam Mat4::to_scale_rotation_translation (column-major)
    t[0] = m[12]; t[1] = m[13]; t[2] = m[14];
    const float det = m[0] * (m[5] * m[10] - m[9] * m[6]) - m[4] * (m[1] * m[10] - m[9] * m[2]) + m[8] * (m[1] * m[6] - m[5] * m[2]);
    const float sign = det < 0.0f ? -1.0f : 1.0f;
    s[0] = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]) * sign;
    s[1] = std::sqrt(m[4] * m[4] + m[5] * m[5] + m[6] * m[6]);
    s[2] = std::sqrt(m[8] * m[8] + m[9] * m[9] + m[10] * m[10]);
    const float is0 = 1.0f / s[0], is1 = 1.0f / s[1], is2 = 1.0f / s[2];
    const float m00 = m[0] * is0, m01 = m[1] * is0, m02 = m[2] * is0, m10 = m[4] * is1, m11 = m[5] * is1, m12 = m[6] * is1, m20 = m[8] * is2, m21 = m[9] * is2, m22 = m[10] * is2;
    // glam Quat::from_rotation_axes
    if (m22 <= 0.0f) {
        const float dif10 = m11 - m00, omm22 = 1.0f - m22;
        if (dif10 <= 0.0f) { const float four_xsq = omm22 - dif10, inv4x = 0.5f / std::sqrt(four_xsq); r[0] = four_xsq * inv4x; r[1] = (m01 + m10) * inv4x; r[2] = (m02 + m20) * inv4x; r[3] = (m12 - m21) * inv4x; }
        else { const float four_ysq = omm22 + dif10, inv4y = 0.5f / std::sqrt(four_ysq); r[0] = (m01 + m10) * inv4y; r[1] = four_ysq * inv4y; r[2] = (m12 + m21) * inv4y; r[3] = (m20 - m02) * inv4y; }
    } else {
        const float sum10 = m11 + m00, opm22 = 1.0f + m22;
        if (sum10 <= 0.0f) { const float four_zsq = opm22 - sum10, inv4z = 0.5f / std::sqrt(four_zsq); r[0] = (m02 + m20) * inv4z; r[1] = (m12 + m21) * inv4z; r[2] = four_zsq * inv4z; r[3] = (m01 - m10) * inv4z; }
        else { const float four_wsq = opm22 + sum10, inv4w = 0.5f / std::sqrt(four_wsq); r[0] = (m12 - m21) * inv4w; r[1] = (m20 - m02) * inv4w; r[2] = (m01 - m10) * inv4w; r[3] = four_wsq * inv4w; }
    }
}

void node_trs(const Value& n, float t[3], float r[4], float s[3]) {   // populate/transforms.rs: node.transform().decomposed()
    const float zero3[3] = {0, 0, 0}, one3[3] = {1, 1, 1}, idq[4] = {0, 0, 0, 1};
    if (n["matrix"].is_array() && n["matrix"].size() == 16) {
        float m[16];
        for (int i = 0; i < 16; i++) m[i] = (float)n["matrix"][(size_t)i].number(i % 5 == 0 ? 1.0 : 0.0);
        mat4_decompose(m, t, r, s);
        return;
    }
    vec_n(n["translation"], t, 3, zero3); vec_n(n["rotation"], r, 4, idq); vec_n(n["scale"], s, 3, one3);
}

bool add_transforms(Loader& L, size_t node, AwsmKey parent) {
    const Value& n = L.doc["nodes"][node];
    if (!n.is_object()) return L.fail("node %zu missing", node);
    if (L.node_keys[node]) return L.fail("node %zu reached twice (not a tree)", node);
    float t[3], r[4], s[3];
    node_trs(n, t, r, s);
    const AwsmKey k = awsm_host_transform_insert(L.h, t, r, s, parent);
    if (!k) return L.fail("node %zu: %s", node, awsm_host_last_error(L.h));
    L.node_keys[node] = k;
    L.info.nodes++;
    const Value& ch = n["children"];
    for (size_t i = 0; i < ch.size(); i++) if (!add_transforms(L, (size_t)ch[i].integer(-1), k)) return false;
    return true;
}

// populate/mesh.rs + gltf/buffers/mesh.rs for one primitive
bool add_primitive(Loader& L, const Value& node, size_t node_index, const Value& mesh, const Value& prim, AwsmKey transform, bool is_joint_node) {
    (void)node_index;
    const bool mesh_on_joint_node = is_joint_node;
    const int64_t mode = prim["mode"].integer(4);
    if (mode != 4 && mode != 5 && mode != 6) return L.fail("primitive mode %lld is not supported (triangles, strips and fans only; buffers/index.rs:203-206)", (long long)mode);
    const Value& attrs = prim["attributes"];
    if (!attrs.has("POSITION")) return L.fail("primitive without POSITION");
    std::vector<float> pos, nrm, tan;
    size_t V = 0;
    if (!read_floats(L, (int)attrs["POSITION"].integer(-1), 3, pos, &V)) return false;
    if (V == 0) return true;
    // indices -> triangle list
    std::vector<uint32_t> raw, idx;
    if (prim.has("indices")) { if (!read_uints(L, (int)prim["indices"].integer(-1), 1, raw, nullptr)) return false; }
    else { raw.resize(V); for (size_t i = 0; i < V; i++) raw[i] = (uint32_t)i; }
    if (mode == 4) { idx = raw; idx.resize(idx.size() / 3 * 3); }
    else if (raw.size() >= 3) {
        for (size_t i = 0; i + 2 < raw.size(); i++) {
            if (mode == 5) { if (i % 2 == 0) { idx.push_back(raw[i]); idx.push_back(raw[i + 1]); idx.push_back(raw[i + 2]); } else { idx.push_back(raw[i]); idx.push_back(raw[i + 2]); idx.push_back(raw[i + 1]); } }
            else { idx.push_back(raw[0]); idx.push_back(raw[i + 1]); idx.push_back(raw[i + 2]); }
        }
    }
    for (uint32_t i : idx) if (i >= V) return L.fail("primitive index %u >= vertex count %zu", i, V);
    if (idx.empty()) return true;
    // attributes
    size_t n;
    if (attrs.has("NORMAL")) { if (!read_floats(L, (int)attrs["NORMAL"].integer(-1), 3, nrm, &n) || n != V) return err_or(L, "NORMAL count differs from POSITION"); }
    else compute_normals(pos, idx, nrm);
    std::vector<std::vector<float>> uvs, colors;
    for (int i = 0; i < 8; i++) {
        const std::string key = "TEXCOORD_" + std::to_string(i);
        if (!attrs.has(key.c_str())) break;
        uvs.emplace_back();
        if (!read_floats(L, (int)attrs[key.c_str()].integer(-1), 2, uvs.back(), &n) || n != V) return err_or(L, "TEXCOORD count differs from POSITION");
    }
    for (int i = 0; i < 4; i++) {
        const std::string key = "COLOR_" + std::to_string(i);
        if (!attrs.has(key.c_str())) break;
        const int acc = (int)attrs[key.c_str()].integer(-1);
        const int comps = type_components(L.doc["accessors"][(size_t)acc]["type"].string());
        std::vector<float> c;
        if (!read_floats(L, acc, comps == 3 ? 3 : 4, c, &n) || n != V) return err_or(L, "COLOR count differs from POSITION");
        colors.emplace_back(V * 4);
        for (size_t v = 0; v < V; v++) {
            float* o = &colors.back()[v * 4];
            if (comps == 3) { o[0] = c[v * 3]; o[1] = c[v * 3 + 1]; o[2] = c[v * 3 + 2]; o[3] = 1.0f; } else memcpy(o, &c[v * 4], 16);
        }
    }
    const int material_index = (int)prim["material"].integer(-1);
    const Value& mat = material_index >= 0 ? L.doc["materials"][(size_t)material_index] : Value();
    if (attrs.has("TANGENT")) { if (!read_floats(L, (int)attrs["TANGENT"].integer(-1), 4, tan, &n) || n != V) return err_or(L, "TANGENT count differs from POSITION"); }
    else if ((mat["normalTexture"].is_object() || mat["extensions"]["KHR_materials_clearcoat"]["clearcoatNormalTexture"].is_object()) && !uvs.empty()) {
        compute_tangents(pos, nrm, uvs[0], idx, tan);                 // ensure_tangents (buffers/tangents.rs:11-98)
        L.info.generated_tangents++;
    }
    // skin sets
    std::vector<std::vector<uint32_t>> joints;
    std::vector<std::vector<float>> weights;
    for (int i = 0; i < 4; i++) {
        const std::string jk = "JOINTS_" + std::to_string(i), wk = "WEIGHTS_" + std::to_string(i);
        if (!attrs.has(jk.c_str()) || !attrs.has(wk.c_str())) break;
        joints.emplace_back(); weights.emplace_back();
        if (!read_uints(L, (int)attrs[jk.c_str()].integer(-1), 4, joints.back(), &n) || n != V) return err_or(L, "JOINTS count differs from POSITION");
        if (!read_floats(L, (int)attrs[wk.c_str()].integer(-1), 4, weights.back(), &n) || n != V) return err_or(L, "WEIGHTS count differs from POSITION");
    }
    // morph targets (buffers/morph.rs): POSITION / NORMAL / TANGENT (vec3) deltas
    const Value& targets = prim["targets"];
    std::vector<std::vector<float>> tp(targets.size()), tn(targets.size()), tt(targets.size());
    std::vector<AwsmHostMorphTarget> mts(targets.size());
    for (size_t t = 0; t < targets.size(); t++) {
        mts[t] = AwsmHostMorphTarget{nullptr, nullptr, nullptr};
        if (targets[t].has("POSITION")) { if (!read_floats(L, (int)targets[t]["POSITION"].integer(-1), 3, tp[t], &n) || n != V) return err_or(L, "morph POSITION count differs"); mts[t].positions = tp[t].data(); }
        if (targets[t].has("NORMAL")) { if (!read_floats(L, (int)targets[t]["NORMAL"].integer(-1), 3, tn[t], &n) || n != V) return err_or(L, "morph NORMAL count differs"); mts[t].normals = tn[t].data(); }
        if (targets[t].has("TANGENT")) { if (!read_floats(L, (int)targets[t]["TANGENT"].integer(-1), 3, tt[t], &n) || n != V) return err_or(L, "morph TANGENT count differs"); mts[t].tangents = tt[t].data(); }
    }
    std::vector<float> morph_weights(targets.size(), 0.0f), animated;
    for (size_t t = 0; t < targets.size() && t < mesh["weights"].size(); t++) morph_weights[t] = (float)mesh["weights"][t].number(0.0);
    if (prim["extras"]["animated_morph_weights"].is_array()) for (size_t t = 0; t < targets.size(); t++) animated.push_back((float)prim["extras"]["animated_morph_weights"][t].number(0.0));
    // skin: one Skins::insert per primitive (populate/mesh.rs)
    AwsmKey skin_key = 0;
    const int64_t skin_index = node["skin"].integer(-1);
    if (skin_index >= 0 && !joints.empty()) {
        const Value& sk = L.doc["skins"][(size_t)skin_index];
        const Value& js = sk["joints"];
        std::vector<AwsmKey> jkeys(js.size());
        for (size_t j = 0; j < js.size(); j++) {
            const size_t jn = (size_t)js[j].integer(-1);
            if (jn >= L.node_keys.size() || !L.node_keys[jn]) return L.fail("skin %lld: joint node %zu is not part of the scene", (long long)skin_index, jn);
            jkeys[j] = L.node_keys[jn];
        }
        std::vector<float> ibm;
        if (sk.has("inverseBindMatrices")) { if (!read_floats(L, (int)sk["inverseBindMatrices"].integer(-1), 16, ibm, &n) || n != js.size()) return err_or(L, "inverseBindMatrices count differs from joints"); }
        else { ibm.assign(js.size() * 16, 0.0f); for (size_t j = 0; j < js.size(); j++) for (int d = 0; d < 4; d++) ibm[j * 16 + d * 5] = 1.0f; }
        std::vector<const uint32_t*> jp; std::vector<const float*> wp;
        for (size_t s2 = 0; s2 < joints.size(); s2++) { jp.push_back(joints[s2].data()); wp.push_back(weights[s2].data()); }
        skin_key = awsm_host_skin_insert(L.h, jkeys.data(), (uint32_t)jkeys.size(), ibm.data(), (uint32_t)joints.size(), jp.data(), wp.data(), (uint32_t)V);
        if (!skin_key) return L.fail("skin %lld: %s", (long long)skin_index, awsm_host_last_error(L.h));
        L.info.skins++;
    }
    AwsmKey mk;
    if (!material_key(L, material_index, &mk)) return false;
    AwsmHostPrimitive p;
    memset(&p, 0, sizeof p);
    p.vertex_count = (uint32_t)V; p.triangle_count = (uint32_t)(idx.size() / 3);
    p.positions = pos.data(); p.normals = nrm.data(); p.tangents = tan.empty() ? nullptr : tan.data(); p.indices = idx.data();
    p.n_uv_sets = (uint32_t)uvs.size(); for (size_t i = 0; i < uvs.size(); i++) p.uv_sets[i] = uvs[i].data();
    p.n_color_sets = (uint32_t)colors.size(); for (size_t i = 0; i < colors.size(); i++) p.color_sets[i] = colors[i].data();
    p.n_morph_targets = (uint32_t)mts.size(); p.morph_targets = mts.empty() ? nullptr : mts.data();
    p.morph_weights = morph_weights.empty() ? nullptr : morph_weights.data();
    p.animated_morph_weights = animated.empty() ? nullptr : animated.data();
    const AwsmKey mesh_key = awsm_host_mesh_insert(L.h, &p, transform, mk, skin_key, 0);
    if (!mesh_key) return L.fail("mesh: %s", awsm_host_last_error(L.h));
    L.info.meshes++; L.info.triangles += p.triangle_count;
    // EXT_mesh_gpu_instancing on the node (gltf/populate/extensions/instancing.rs:9-160): TRANSLATION / SCALE are float VEC3, ROTATION is a
    // VEC4 of floats or of integers cast as they are (`v as f32`: the reference does not normalise them); the count is that of the first
    // attribute present, missing ones are identity; every mesh that hangs on the node's own transform becomes instanced (a skinned mesh
    // on a joint node has a transform of its own, populate/mesh.rs:36-52, and stays single).
    const Value& ext_inst = node["extensions"]["EXT_mesh_gpu_instancing"]["attributes"];
    if (ext_inst.is_object() && !mesh_on_joint_node) {
        auto attr = [&](const char* upper, const char* lower) -> const Value& { return ext_inst.has(lower) ? ext_inst[lower] : ext_inst[upper]; };
        std::vector<float> tr, ro, sc;
        size_t nt = 0, nr = 0, ns = 0;
        const Value &at = attr("TRANSLATION", "translation"), &ar = attr("ROTATION", "rotation"), &as = attr("SCALE", "scale");
        const auto is_f32 = [&](const Value& a) { return L.doc["accessors"][(size_t)a.integer(-1)]["componentType"].integer(0) == 5126; };
        if (at.is_number()) { if (!is_f32(at)) return L.fail("EXT_mesh_gpu_instancing: translation isn't a Vec3F32"); if (!read_floats(L, (int)at.integer(-1), 3, tr, &nt)) return false; }
        if (ar.is_number() && !read_floats(L, (int)ar.integer(-1), 4, ro, &nr, true)) return false;
        if (as.is_number()) { if (!is_f32(as)) return L.fail("EXT_mesh_gpu_instancing: scale isn't a Vec3F32"); if (!read_floats(L, (int)as.integer(-1), 3, sc, &ns)) return false; }
        const size_t count = at.is_number() ? nt : (ar.is_number() ? nr : (as.is_number() ? ns : 0));
        if ((ar.is_number() && nr < count) || (as.is_number() && ns < count)) return L.fail("EXT_mesh_gpu_instancing: attribute shorter than the instance count %zu", count);
        if (count) {
            std::vector<float> trs(count * 10);
            for (size_t i = 0; i < count; i++) {
                float* o = &trs[i * 10];
                for (int c = 0; c < 3; c++) o[c] = at.is_number() ? tr[i * 3 + c] : 0.0f;
                for (int c = 0; c < 4; c++) o[3 + c] = ar.is_number() ? ro[i * 4 + c] : (c == 3 ? 1.0f : 0.0f);
                for (int c = 0; c < 3; c++) o[7 + c] = as.is_number() ? sc[i * 3 + c] : 1.0f;
            }
            if (awsm_host_mesh_set_instances(L.h, mesh_key, trs.data(), (uint32_t)count)) return L.fail("EXT_mesh_gpu_instancing: %s", awsm_host_last_error(L.h));
            L.info.instanced_meshes++;
        }
    }
    // the same carried per primitive by scenes this repo exports (a SceneDesc can instance one primitive of a node): extras.instances = [[t3, r4, s3], ...]
    const Value& inst = prim["extras"]["instances"];
    if (inst.is_array() && inst.size()) {
        std::vector<float> trs(inst.size() * 10);
        for (size_t i = 0; i < inst.size(); i++) for (int c = 0; c < 10; c++) trs[i * 10 + c] = (float)inst[i][(size_t)c].number(c == 6 ? 1.0 : (c >= 7 ? 1.0 : 0.0));
        if (awsm_host_mesh_set_instances(L.h, mesh_key, trs.data(), (uint32_t)inst.size())) return L.fail("instances: %s", awsm_host_last_error(L.h));
    }
    return true;
}

bool add_meshes(Loader& L, size_t node, const std::vector<bool>& is_joint) {
    const Value& n = L.doc["nodes"][node];
    const int64_t mi = n["mesh"].integer(-1);
    if (mi >= 0) {
        const Value& mesh = L.doc["meshes"][(size_t)mi];
        if (!mesh.is_object()) return L.fail("node %zu: mesh %lld missing", node, (long long)mi);
        AwsmKey tk = L.node_keys[node];
        if (is_joint[node]) {   // populate/mesh.rs:36-52: a mesh on a joint node gets a fresh identity transform under the joint's parent
            const float t[3] = {0, 0, 0}, r[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
            tk = awsm_host_transform_insert(L.h, t, r, s, awsm_host_transform_parent(L.h, tk));
            if (!tk) return L.fail("node %zu: %s", node, awsm_host_last_error(L.h));
        }
        const Value& prims = mesh["primitives"];
        for (size_t p = 0; p < prims.size(); p++) if (!add_primitive(L, n, node, mesh, prims[p], tk, is_joint[node])) return false;
    }
    const Value& ch = n["children"];
    for (size_t i = 0; i < ch.size(); i++) if (!add_meshes(L, (size_t)ch[i].integer(-1), is_joint)) return false;
    return true;
}

// KHR_lights_punctual: node.extensions.KHR_lights_punctual.light -> lights.rs records, in node order
bool add_lights(Loader& L) {
    const Value& defs = L.doc["extensions"]["KHR_lights_punctual"]["lights"];
    if (!defs.is_array()) return true;
    const Value& nodes = L.doc["nodes"];
    for (size_t i = 0; i < nodes.size(); i++) {
        const Value& ref = nodes[i]["extensions"]["KHR_lights_punctual"]["light"];
        if (!ref.is_number()) continue;
        const Value& d = defs[(size_t)ref.integer(-1)];
        if (!d.is_object()) continue;
        if (!L.node_keys[i] && !d["extras"].has("position")) continue;     // a light on a node outside the scene: only with explicit vectors
        float w[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (L.node_keys[i] && awsm_host_transform_world(L.h, L.node_keys[i], w)) return L.fail("light node %zu: %s", i, awsm_host_last_error(L.h));
        AwsmHostLight l;
        memset(&l, 0, sizeof l);
        const std::string& ty = d["type"].string();
        l.kind = ty == "directional" ? 1u : (ty == "point" ? 2u : 3u);
        const float one3[3] = {1, 1, 1};
        vec_n(d["color"], l.color, 3, one3);
        l.intensity = (float)d["intensity"].number(1.0);
        l.range = (float)d["range"].number(0.0);
        l.position[0] = w[12]; l.position[1] = w[13]; l.position[2] = w[14];
        l.direction[0] = -w[8]; l.direction[1] = -w[9]; l.direction[2] = -w[10];          // a light points down its node's -Z
        l.inner_angle = std::cos((float)d["spot"]["innerConeAngle"].number(0.0));
        l.outer_angle = std::cos((float)d["spot"]["outerConeAngle"].number(0.7853981633974483));
        if (d["extras"].has("direction")) { const float z3[3] = {0, 0, -1}; vec_n(d["extras"]["direction"], l.direction, 3, z3); }   // scenes exported by this repo keep the exact values
        if (d["extras"].has("position")) { const float z3[3] = {0, 0, 0}; vec_n(d["extras"]["position"], l.position, 3, z3); }
        if (d["extras"].has("inner_angle")) l.inner_angle = (float)d["extras"]["inner_angle"].number(l.inner_angle);
        if (d["extras"].has("outer_angle")) l.outer_angle = (float)d["extras"]["outer_angle"].number(l.outer_angle);
        if (!awsm_host_light_insert(L.h, &l)) return L.fail("light: %s", awsm_host_last_error(L.h));
        L.info.lights++;
    }
    return true;
}

bool load(Loader& L, const char* path, int scene_index) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return L.fail("cannot read '%s'", path);
    const std::string p = path;
    const size_t slash = p.find_last_of('/');
    L.dir = slash == std::string::npos ? std::string() : p.substr(0, slash + 1);
    const uint8_t* json = file.data();
    size_t json_len = file.size();
    const uint8_t* bin = nullptr;
    size_t bin_len = 0;
    if (file.size() >= 12 && memcmp(file.data(), "glTF", 4) == 0) {                 // GLB container (loader.rs)
        uint32_t version, total;
        memcpy(&version, file.data() + 4, 4); memcpy(&total, file.data() + 8, 4);
        if (version != 2 || total > file.size()) return L.fail("GLB header: version %u, length %u of %zu", version, total, file.size());
        size_t pos = 12;
        json = nullptr;
        while (pos + 8 <= total) {
            uint32_t clen, ctype;
            memcpy(&clen, file.data() + pos, 4); memcpy(&ctype, file.data() + pos + 4, 4);
            if (pos + 8 + (size_t)clen > total) return L.fail("GLB chunk exceeds the file");
            if (ctype == 0x4E4F534Au && !json) { json = file.data() + pos + 8; json_len = clen; }
            else if (ctype == 0x004E4942u && !bin) { bin = file.data() + pos + 8; bin_len = clen; }
            pos += 8 + (size_t)clen + ((4 - (clen & 3)) & 3);
        }
        if (!json) return L.fail("GLB without a JSON chunk");
    }
    std::string jerr;
    if (!awsm_json::parse(reinterpret_cast<const char*>(json), json_len, L.doc, jerr)) return L.fail("JSON: %s", jerr.c_str());
    if (L.doc["asset"]["version"].string().compare(0, 1, "2") != 0) return L.fail("asset.version '%s' (glTF 2.x required)", L.doc["asset"]["version"].string().c_str());
    const Value& req = L.doc["extensionsRequired"];
    static const char* known[] = {"KHR_materials_emissive_strength", "KHR_materials_ior", "KHR_materials_specular", "KHR_materials_transmission", "KHR_materials_volume",
                                  "KHR_materials_clearcoat", "KHR_materials_sheen", "KHR_materials_unlit", "KHR_texture_transform", "KHR_lights_punctual"};
    for (size_t i = 0; i < req.size(); i++) {
        bool ok = false;
        for (const char* k : known) ok = ok || req[i].string() == k;
        if (!ok) return L.fail("required extension '%s' is not supported", req[i].string().c_str());
    }
    const Value& bufs = L.doc["buffers"];
    L.buffers.resize(bufs.size());
    for (size_t i = 0; i < bufs.size(); i++) {
        if (bufs[i].has("uri")) { if (!load_uri(L, bufs[i]["uri"].string(), L.buffers[i])) return false; }
        else if (i == 0 && bin) L.buffers[i].assign(bin, bin + bin_len);
        else return L.fail("buffer %zu has no uri and there is no GLB binary chunk", i);
        if (L.buffers[i].size() < (size_t)bufs[i]["byteLength"].integer(0)) return L.fail("buffer %zu is shorter than its byteLength", i);
    }
    // images (decoded, with the mip kind their first role implies), samplers, then the reference's populate order
    if (!load_images(L) || !load_samplers(L)) return false;
    const Value& scenes = L.doc["scenes"];
    const int64_t si = scene_index >= 0 ? scene_index : L.doc["scene"].integer(0);
    if (si < 0 || (size_t)si >= scenes.size()) return L.fail("scene %lld missing (populate.rs:168-183)", (long long)si);
    const Value& roots = scenes[(size_t)si]["nodes"];
    const size_t n_nodes = L.doc["nodes"].size();
    L.node_keys.assign(n_nodes, 0);
    for (size_t i = 0; i < roots.size(); i++) { const size_t r = (size_t)roots[i].integer(-1); if (r >= n_nodes || !add_transforms(L, r, 0)) return L.err.empty() ? L.fail("scene root %zu missing", r) : false; }
    std::vector<bool> is_joint(n_nodes, false);
    const Value& skins = L.doc["skins"];
    for (size_t s = 0; s < skins.size(); s++) for (size_t j = 0; j < skins[s]["joints"].size(); j++) { const size_t jn = (size_t)skins[s]["joints"][j].integer(-1); if (jn < n_nodes) is_joint[jn] = true; }
    for (size_t i = 0; i < roots.size(); i++) if (!add_meshes(L, (size_t)roots[i].integer(-1), is_joint)) return false;
    if (awsm_host_update_transforms(L.h)) return L.fail("update_transforms: %s", awsm_host_last_error(L.h));
    return add_lights(L);
}

bool err_or(Loader& L, const char* msg) { if (L.err.empty()) L.fail("%s", msg); return false; }

}  // namespace

extern "C" int awsm_host_decode_image(const uint8_t* data, size_t len, uint8_t* rgba_out, size_t cap, uint32_t* width, uint32_t* height, char* err_out, size_t err_cap) {
    if (!data || !width || !height) return AWSM_ERR_INVALID_ARGUMENT;
    std::vector<uint8_t> rgba;
    std::string err;
    bool ok = false;
    if (awsm_png::is_png(data, len)) ok = awsm_png::decode(data, len, rgba, *width, *height, err);
    else if (awsm_jpeg::is_jpeg(data, len)) ok = awsm_jpeg::decode(data, len, rgba, *width, *height, err);
    else err = "this image format is not supported (PNG and baseline JPEG only)";
    if (!ok) { if (err_out && err_cap) snprintf(err_out, err_cap, "%s", err.c_str()); return err.find("not supported") != std::string::npos ? AWSM_ERR_UNSUPPORTED : AWSM_ERR_INVALID_ARGUMENT; }
    if (rgba_out) { if (cap < rgba.size()) return AWSM_ERR_OUT_OF_RANGE; memcpy(rgba_out, rgba.data(), rgba.size()); }
    return AWSM_OK;
}

extern "C" int awsm_host_load_gltf(AwsmHost* h, const char* path, int scene_index, AwsmGltfInfo* info_out, char* err_out, size_t err_cap) {
    if (!h || !path) return AWSM_ERR_INVALID_ARGUMENT;
    Loader L;
    L.h = h;
    const bool ok = load(L, path, scene_index);
    if (info_out) *info_out = L.info;
    if (!ok) {
        if (err_out && err_cap) { snprintf(err_out, err_cap, "%s", L.err.c_str()); }
        return L.err.find("not supported") != std::string::npos ? AWSM_ERR_UNSUPPORTED : AWSM_ERR_INVALID_ARGUMENT;
    }
    if (err_out && err_cap) err_out[0] = 0;
    return AWSM_OK;
}
