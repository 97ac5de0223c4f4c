// ply_io.cpp — native PLY reader/writer behind include/gsx.h (gsx_ply_*).
//
// Replaces the plyfile calls of the reference: PlyData.read (deep_learning_segmentation.py:29,
// ply_handler.py:44-45) and PlyData([...], text=False).write (deep_learning_segmentation.py:331-332,
// ply_handler.py:35-37).  The header grammar written here is the one the reference viewer parses
// (gaussians_selection.js:466-500): "end_header\n" within the first 10 KiB, "element vertex <n>\n",
// "property <type> <name>" lines, little-endian rows.
// Binary files are mmap'ed privately (copy-on-write): rows are used in place, edits never touch the
// source file.  ASCII files are parsed into the same row layout.  Only the vertex element is kept,
// like save_labeled_ply (deep_learning_segmentation.py:331).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gsx.h"

// error plumbing of libgsx (gsx_api.cpp); declared here so that this file needs no HIP header and can be
// built host-only (tests/test_ply_asan.py runs it under AddressSanitizer)
namespace gsx {
struct Ctx;
int fail(Ctx* c, int code, const char* fmt, ...);
}  // namespace gsx

namespace {

struct Prop {
    std::string name;
    int type;  // gsx_ply_type
    int64_t offset;
};

const int kTypeSize[8] = {1, 1, 2, 2, 4, 4, 4, 8};
const char* const kTypeName[8] = {"char", "uchar", "short", "ushort", "int", "uint", "float", "double"};

int parse_type(const std::string& t) {
    static const char* const alias[8][3] = {{"char", "int8", ""},     {"uchar", "uint8", ""},   {"short", "int16", ""},
                                            {"ushort", "uint16", ""}, {"int", "int32", ""},     {"uint", "uint32", ""},
                                            {"float", "float32", ""}, {"double", "float64", ""}};
    for (int k = 0; k < 8; ++k)
        for (int a = 0; a < 2; ++a)
            if (t == alias[k][a]) return k;
    return -1;
}

}  // namespace

struct gsx_ply {
    int64_t n = 0;
    int64_t stride = 0;
    std::vector<Prop> props;
    // storage: either a private mapping of the file or an owned buffer
    void* map = nullptr;
    size_t map_len = 0;
    std::vector<unsigned char> owned;
    unsigned char* rows = nullptr;
    std::string path;
};

namespace {

double load_as_double(const unsigned char* p, int type) {
    switch (type) {
        case GSX_PLY_CHAR: { int8_t v; std::memcpy(&v, p, 1); return v; }
        case GSX_PLY_UCHAR: { uint8_t v; std::memcpy(&v, p, 1); return v; }
        case GSX_PLY_SHORT: { int16_t v; std::memcpy(&v, p, 2); return v; }
        case GSX_PLY_USHORT: { uint16_t v; std::memcpy(&v, p, 2); return v; }
        case GSX_PLY_INT: { int32_t v; std::memcpy(&v, p, 4); return v; }
        case GSX_PLY_UINT: { uint32_t v; std::memcpy(&v, p, 4); return v; }
        case GSX_PLY_FLOAT: { float v; std::memcpy(&v, p, 4); return v; }
        default: { double v; std::memcpy(&v, p, 8); return v; }
    }
}

void store_from_double(unsigned char* p, int type, double d) {
    switch (type) {
        case GSX_PLY_CHAR: { int8_t v = (int8_t)d; std::memcpy(p, &v, 1); break; }
        case GSX_PLY_UCHAR: { uint8_t v = (uint8_t)d; std::memcpy(p, &v, 1); break; }
        case GSX_PLY_SHORT: { int16_t v = (int16_t)d; std::memcpy(p, &v, 2); break; }
        case GSX_PLY_USHORT: { uint16_t v = (uint16_t)d; std::memcpy(p, &v, 2); break; }
        case GSX_PLY_INT: { int32_t v = (int32_t)d; std::memcpy(p, &v, 4); break; }
        case GSX_PLY_UINT: { uint32_t v = (uint32_t)d; std::memcpy(p, &v, 4); break; }
        case GSX_PLY_FLOAT: { float v = (float)d; std::memcpy(p, &v, 4); break; }
        default: std::memcpy(p, &d, 8);
    }
}

template <class F>
void parallel_rows(int64_t n, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, n / 65536));
    if (nt <= 1) {
        f((int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([=] { f(n * t / nt, n * (t + 1) / nt); });
    for (auto& t : th) t.join();
}

const Prop* find_prop(const gsx_ply* p, const char* name) {
    for (const auto& q : p->props)
        if (q.name == name) return &q;
    return nullptr;
}

}  // namespace

namespace {
int ply_open_impl(const char* path, gsx_ply** out);
int ply_write_impl(const gsx_ply* p, const char* path, const int32_t* labels, int32_t text);

// no exception may cross the C boundary (std::bad_alloc from a vector sized by a hostile header, std::system_error
// from a thread that cannot be started, ...)
template <class F>
int guarded(const char* who, F f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return gsx::fail(nullptr, GSX_E_IO, "%s: out of memory", who);
    } catch (const std::exception& e) {
        return gsx::fail(nullptr, GSX_E_IO, "%s: %s", who, e.what());
    } catch (...) {
        return gsx::fail(nullptr, GSX_E_IO, "%s: unknown failure", who);
    }
}
}  // namespace

extern "C" {

int gsx_ply_open(const char* path, gsx_ply** out) {
    if (!path || !out) return gsx::fail(nullptr, GSX_E_INVALID, "ply_open: NULL argument");
    *out = nullptr;
    return guarded("ply_open", [&] { return ply_open_impl(path, out); });
}

}  // extern "C"

namespace {
int ply_open_impl(const char* path, gsx_ply** out) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return gsx::fail(nullptr, GSX_E_IO, "ply_open: cannot open %s: %s", path, std::strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 4) {
        ::close(fd);
        return gsx::fail(nullptr, GSX_E_IO, "ply_open: %s is empty or unreadable", path);
    }
    const size_t len = (size_t)st.st_size;
    void* map = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) return gsx::fail(nullptr, GSX_E_IO, "ply_open: mmap of %s failed: %s", path, std::strerror(errno));
    const char* txt = static_cast<const char*>(map);
    auto bail = [&](const char* msg) {
        munmap(map, len);
        return gsx::fail(nullptr, GSX_E_IO, "ply_open: %s: %s", path, msg);
    };
    if (std::memcmp(txt, "ply", 3) != 0) return bail("not a PLY file");
    const std::string marker = "end_header\n";
    const size_t scan = std::min(len, (size_t)1 << 20);
    const char* eh = static_cast<const char*>(memmem(txt, scan, marker.data(), marker.size()));
    if (!eh) return bail("no end_header within the first MiB");
    const size_t data_off = (size_t)(eh - txt) + marker.size();
    std::string header(txt, (size_t)(eh - txt));
    gsx_ply* p = new gsx_ply();
    p->path = path;
    int format = -1;  // 0 ascii, 1 binary LE
    bool in_vertex = false, seen_vertex = false, after_vertex = false;
    size_t pos = 0;
    while (pos < header.size()) {
        size_t nl = header.find('\n', pos);
        if (nl == std::string::npos) nl = header.size();
        std::string line = header.substr(pos, nl - pos);
        pos = nl + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        char a[64] = {0}, b[64] = {0}, c[128] = {0};
        const int k = std::sscanf(line.c_str(), "%63s %63s %127s", a, b, c);
        if (k < 1) continue;
        const std::string kw(a);
        if (kw == "format") {
            if (std::string(b) == "ascii") format = 0;
            else if (std::string(b) == "binary_little_endian") format = 1;
            else {
                delete p;
                return bail("only ascii and binary_little_endian are supported");
            }
        } else if (kw == "element") {
            if (std::string(b) == "vertex") {
                char* endp = nullptr;
                errno = 0;
                const long long cnt = std::strtoll(c, &endp, 10);
                if (endp == c || *endp != 0 || errno == ERANGE || cnt < 0) {
                    delete p;
                    return bail("`element vertex` carries no valid count");
                }
                p->n = cnt;
                in_vertex = true;
                seen_vertex = true;
            } else {
                if (in_vertex) after_vertex = true;
                if (!seen_vertex) {
                    delete p;
                    return bail("an element precedes `vertex`: unsupported layout");
                }
                in_vertex = false;
            }
        } else if (kw == "property" && in_vertex) {
            if (std::string(b) == "list") {
                delete p;
                return bail("list properties in the vertex element are unsupported");
            }
            const int t = parse_type(b);
            if (t < 0 || k < 3) {
                delete p;
                return bail("unknown property type");
            }
            p->props.push_back({c, t, p->stride});
            p->stride += kTypeSize[t];
        }
    }
    (void)after_vertex;  // later elements (faces, ...) are ignored, as save_labeled_ply drops them
    if (!seen_vertex || format < 0 || p->n < 0 || p->props.empty()) {
        delete p;
        return bail("header lacks format / element vertex / properties");
    }
    // sizes in the divided form: a header may promise any count, n * stride must never be formed before it is known to fit
    if (p->stride <= 0 || (uint64_t)p->n > (uint64_t)(len - data_off) / (uint64_t)(format == 1 ? p->stride : 1)) {
        // binary: n rows of `stride` bytes must be there; ascii: every value needs at least one character
        delete p;
        return bail("file is shorter than its header promises");
    }
    if (format == 1) {
        p->map = map;
        p->map_len = len;
        p->rows = static_cast<unsigned char*>(map) + data_off;
    } else {
        if ((uint64_t)p->n > (uint64_t)PTRDIFF_MAX / (uint64_t)p->stride / 2) {
            delete p;
            return bail("vertex count too large");
        }
        p->owned.resize((size_t)p->n * (size_t)p->stride);
        p->rows = p->owned.data();
        const char* cur = txt + data_off;
        const char* end = txt + len;
        std::string tok;
        for (int64_t i = 0; i < p->n; ++i)
            for (const auto& q : p->props) {
                while (cur < end && (*cur == ' ' || *cur == '\n' || *cur == '\r' || *cur == '\t')) ++cur;
                const char* s = cur;
                while (cur < end && !(*cur == ' ' || *cur == '\n' || *cur == '\r' || *cur == '\t')) ++cur;
                if (s == cur) {
                    munmap(map, len);
                    delete p;
                    return gsx::fail(nullptr, GSX_E_IO, "ply_open: %s: ascii body ends early", path);
                }
                tok.assign(s, cur);
                store_from_double(p->rows + i * p->stride + q.offset, q.type, std::strtod(tok.c_str(), nullptr));
            }
        munmap(map, len);
    }
    *out = p;
    return GSX_OK;
}
}  // namespace

extern "C" {

void gsx_ply_close(gsx_ply* p) {
    if (!p) return;
    if (p->map) munmap(p->map, p->map_len);
    delete p;
}

int64_t gsx_ply_num_vertices(const gsx_ply* p) { return p ? p->n : 0; }
int32_t gsx_ply_num_properties(const gsx_ply* p) { return p ? (int32_t)p->props.size() : 0; }
int64_t gsx_ply_row_stride(const gsx_ply* p) { return p ? p->stride : 0; }
void* gsx_ply_rows(gsx_ply* p) { return p ? p->rows : nullptr; }

int gsx_ply_property(const gsx_ply* p, int32_t i, const char** name, int32_t* type, int64_t* offset) {
    if (!p || i < 0 || i >= (int32_t)p->props.size()) return gsx::fail(nullptr, GSX_E_INVALID, "ply_property: index out of range");
    if (name) *name = p->props[i].name.c_str();
    if (type) *type = p->props[i].type;
    if (offset) *offset = p->props[i].offset;
    return GSX_OK;
}

int gsx_ply_read_f32(const gsx_ply* p, const char* name, float* out) {
    if (!p || !name || (!out && p->n > 0)) return gsx::fail(nullptr, GSX_E_INVALID, "ply_read_f32: NULL argument");
    const Prop* q = find_prop(p, name);
    if (!q) return gsx::fail(nullptr, GSX_E_INVALID, "ply_read_f32: no property '%s'", name);
    const unsigned char* base = p->rows + q->offset;
    const int64_t stride = p->stride;
    const int type = q->type;
    const int64_t n = p->n;
    return guarded("ply_read_f32", [&] {
        parallel_rows(n, [=](int64_t lo, int64_t hi) {
            if (type == GSX_PLY_FLOAT)
                for (int64_t i = lo; i < hi; ++i) std::memcpy(out + i, base + i * stride, 4);
            else
                for (int64_t i = lo; i < hi; ++i) out[i] = (float)load_as_double(base + i * stride, type);
        });
        return (int)GSX_OK;
    });
}

int gsx_ply_set_f32(gsx_ply* p, const char* name, const float* in) {
    if (!p || !name || (!in && p->n > 0)) return gsx::fail(nullptr, GSX_E_INVALID, "ply_set_f32: NULL argument");
    const Prop* q = find_prop(p, name);
    if (!q) return gsx::fail(nullptr, GSX_E_INVALID, "ply_set_f32: no property '%s'", name);
    unsigned char* base = p->rows + q->offset;
    const int64_t stride = p->stride;
    const int type = q->type;
    const int64_t n = p->n;
    return guarded("ply_set_f32", [&] {
        parallel_rows(n, [=](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) store_from_double(base + i * stride, type, (double)in[i]);
        });
        return (int)GSX_OK;
    });
}

// All vertex properties (+ a trailing `property int label` when labels != NULL), vertex element only.
int gsx_ply_write(const gsx_ply* p, const char* path, const int32_t* labels, int32_t text) {
    if (!p || !path) return gsx::fail(nullptr, GSX_E_INVALID, "ply_write: NULL argument");
    if (labels && find_prop(p, "label"))
        return gsx::fail(nullptr, GSX_E_INVALID, "ply_write: the vertex element already has a 'label' property");
    return guarded("ply_write", [&] { return ply_write_impl(p, path, labels, text); });
}

}  // extern "C"

namespace {
// The rows may be a private mapping of the very file being written (--output_file == --ply_file works with plyfile,
// which holds the data in memory): truncating it in place would pull the pages from under the mapping (SIGBUS) and
// destroy the input.  So the file is always written next to its destination and renamed over it when complete; a
// failed write leaves the destination untouched.
int ply_write_impl(const gsx_ply* p, const char* path, const int32_t* labels, int32_t text) {
    const std::string tmp = std::string(path) + ".gsx-tmp-" + std::to_string((long long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return gsx::fail(nullptr, GSX_E_IO, "ply_write: cannot create %s: %s", tmp.c_str(), std::strerror(errno));
    std::string h = "ply\nformat ";
    h += text ? "ascii 1.0\n" : "binary_little_endian 1.0\n";
    h += "element vertex " + std::to_string(p->n) + "\n";
    for (const auto& q : p->props) h += std::string("property ") + kTypeName[q.type] + " " + q.name + "\n";
    if (labels) h += "property int label\n";
    h += "end_header\n";
    bool ok = std::fwrite(h.data(), 1, h.size(), f) == h.size();
    if (!text) {
        if (!labels) {
            ok = ok && std::fwrite(p->rows, (size_t)p->stride, (size_t)p->n, f) == (size_t)p->n;
        } else {
            const int64_t os = p->stride + 4;
            const int64_t chunk = 1 << 16;
            std::vector<unsigned char> buf((size_t)(os * chunk));
            for (int64_t i0 = 0; i0 < p->n && ok; i0 += chunk) {
                const int64_t m = std::min(chunk, p->n - i0);
                const unsigned char* src = p->rows + i0 * p->stride;
                const int64_t stride = p->stride;
                unsigned char* dst = buf.data();
                const int32_t* lab = labels + i0;
                parallel_rows(m, [=](int64_t lo, int64_t hi) {
                    for (int64_t i = lo; i < hi; ++i) {
                        std::memcpy(dst + i * os, src + i * stride, (size_t)stride);
                        std::memcpy(dst + i * os + stride, lab + i, 4);
                    }
                });
                ok = std::fwrite(buf.data(), (size_t)os, (size_t)m, f) == (size_t)m;
            }
        }
    } else {
        char num[64];
        std::string line;
        for (int64_t i = 0; i < p->n && ok; ++i) {
            line.clear();
            for (const auto& q : p->props) {
                const unsigned char* s = p->rows + i * p->stride + q.offset;
                if (q.type == GSX_PLY_FLOAT) {
                    float v;
                    std::memcpy(&v, s, 4);
                    std::snprintf(num, sizeof num, "%.9g", (double)v);
                } else if (q.type == GSX_PLY_DOUBLE) {
                    std::snprintf(num, sizeof num, "%.17g", load_as_double(s, q.type));
                } else {
                    std::snprintf(num, sizeof num, "%lld", (long long)load_as_double(s, q.type));
                }
                if (!line.empty()) line += ' ';
                line += num;
            }
            if (labels) line += " " + std::to_string(labels[i]);
            line += '\n';
            ok = std::fwrite(line.data(), 1, line.size(), f) == line.size();
        }
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) {
        std::remove(tmp.c_str());
        return gsx::fail(nullptr, GSX_E_IO, "ply_write: short write to %s", tmp.c_str());
    }
    if (std::rename(tmp.c_str(), path) != 0) {
        const int e = errno;
        std::remove(tmp.c_str());
        return gsx::fail(nullptr, GSX_E_IO, "ply_write: cannot move the finished file to %s: %s", path, std::strerror(e));
    }
    return GSX_OK;
}
}  // namespace
