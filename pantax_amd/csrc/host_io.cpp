// host_io.cpp -- see host_io.hpp.  Host-only C++17; no device code.
#include <dlfcn.h>
#include "host_io.hpp"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <charconv>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>

namespace ptx {

static std::vector<std::string> split_tab(const std::string &line) {
    std::vector<std::string> out;
    size_t b = 0;
    while (true) {
        size_t e = line.find('\t', b);
        if (e == std::string::npos) { out.push_back(line.substr(b)); break; }
        out.push_back(line.substr(b, e - b));
        b = e + 1;
    }
    return out;
}
static void rstrip(std::string &s) { while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back(); }

std::string read_species_range(const std::string &path, std::vector<RangeRow> &out) {
    std::ifstream f(path);
    if (!f) return "cannot open species range file " + path;
    std::string line;
    while (std::getline(f, line)) {
        rstrip(line);
        if (line.empty()) continue;
        auto p = split_tab(line);
        if (p.size() < 3) return "malformed species range line: " + line;
        RangeRow r;
        r.species = p[0];
        try { r.start = std::stoll(p[1]); r.end = std::stoll(p[2]); r.is_pan = p.size() > 3 ? std::stoi(p[3]) : 0; }
        catch (...) { return "malformed species range line: " + line; }
        out.push_back(r);
    }
    return "";
}

std::string read_species_len(const std::string &path, std::vector<std::pair<std::string, double>> &out) {
    std::ifstream f(path);
    if (!f) return "cannot open species length file " + path;
    std::string line;
    while (std::getline(f, line)) {
        rstrip(line);
        if (line.empty()) continue;
        auto p = split_tab(line);
        if (p.size() < 2) return "malformed species length line: " + line;
        try { out.emplace_back(p[0], std::stod(p[1])); } catch (...) { return "malformed species length line: " + line; }
    }
    return "";
}

// profile.rs:3105-3146: hap_id = first two '_' tokens of Path(id).file_stem()
static std::string hap_id_of(const std::string &id) {
    size_t slash = id.find_last_of('/');
    std::string name = slash == std::string::npos ? id : id.substr(slash + 1);
    size_t dot = name.find_last_of('.');
    std::string stem = (dot == std::string::npos || dot == 0) ? name : name.substr(0, dot);
    size_t u1 = stem.find('_');
    if (u1 == std::string::npos) return stem;
    size_t u2 = stem.find('_', u1 + 1);
    return u2 == std::string::npos ? stem : stem.substr(0, u2);
}

std::string read_genomes_info(const std::string &path, std::vector<GenomeRow> &out) {
    std::ifstream f(path);
    if (!f) return "cannot open genomes metadata file " + path;
    std::string line;
    bool header = true;
    while (std::getline(f, line)) {
        rstrip(line);
        if (header) { header = false; continue; }
        if (line.empty()) continue;
        auto p = split_tab(line);
        if (p.size() < 5) return "malformed genomes_info line: " + line;
        out.push_back({p[0], p[1], p[2], hap_id_of(p[4])});
    }
    return "";
}

static void finish_graph(std::map<std::string, std::vector<uint32_t>> &paths, HostGraph &g) {
    g.hap_names.clear(); g.path_off.assign(1, 0); g.path_nodes.clear();
    for (auto &kv : paths) {   // std::map<std::string> iterates in byte-wise order, like BTreeMap<String,_>
        g.hap_names.push_back(kv.first);
        g.path_nodes.insert(g.path_nodes.end(), kv.second.begin(), kv.second.end());
        g.path_off.push_back(g.path_nodes.size());
    }
}

std::string read_gfa(const std::string &path, HostGraph &g) {
    std::ifstream f(path);
    if (!f) return "cannot open GFA " + path;
    g.node_len.clear();
    std::map<std::string, std::vector<uint32_t>> paths;
    std::string line;
    size_t node_index = 0;
    while (std::getline(f, line)) {
        rstrip(line);
        if (line.empty()) continue;
        if (line[0] == 'S') {                                  // profile.rs:481-495
            auto p = split_tab(line);
            if (p.size() < 3) continue;
            size_t id = 0;
            try { id = std::stoull(p[1]); } catch (...) { return "GFA S line with a non-numeric id: " + p[1]; }
            if (id != node_index + 1) return "GFA node ids must be dense and ordered (profile.rs:489): got " + p[1];
            ++node_index;
            if (p[2].empty()) return "GFA node of length 0 (profile.rs:494)";
            g.node_len.push_back((int64_t)p[2].size());
        } else if (line[0] == 'W' || line[0] == 'P') {         // profile.rs:496-541
            auto p = split_tab(line);
            std::string hap;
            const std::string *field;
            if (p[0] == "W") { if (p.size() < 2) continue; hap = p[1]; field = &p.back(); }
            else { if (p.size() < 3) continue; hap = p[1].substr(0, p[1].find('#')); field = &p[2]; }
            std::vector<uint32_t> ids;
            const std::string &s = *field;
            for (size_t i = 0; i < s.size();) {
                if (s[i] >= '0' && s[i] <= '9') {
                    if (p[0] == "W" && i > 0 && s[i - 1] == '-') return "negative node id in W line (regex -?\\d+ would wrap in the reference)";
                    uint64_t v = 0;
                    while (i < s.size() && s[i] >= '0' && s[i] <= '9') { v = v * 10 + (uint64_t)(s[i] - '0'); ++i; }
                    if (v == 0 || v > 0xFFFFFFFFull) return "node id out of range in path of " + hap;
                    ids.push_back((uint32_t)(v - 1));
                } else ++i;
            }
            auto &dst = paths[hap];                            // contigs of one haplotype are concatenated (:540)
            dst.insert(dst.end(), ids.begin(), ids.end());
        }
    }
    finish_graph(paths, g);
    return "";
}

// bincode-1 (fixed-int, little endian) image of `Graph { nodes_len: Vec<i64>, paths: BTreeMap<String, Vec<usize>> }`
// (types.rs:51-55, zip.rs:171-190): u64 len + i64s; u64 map len; per entry u64 key len + bytes + u64 vec len + u64s
static std::string parse_graph_bincode(const char *p, size_t n, const std::string &path, HostGraph &g) {
    size_t off = 0;
    auto rd64 = [&](uint64_t &v) { if (off + 8 > n) return false; std::memcpy(&v, p + off, 8); off += 8; return true; };
    uint64_t nn = 0;
    if (!rd64(nn) || nn > (n - off) / 8) return "truncated graph file " + path;
    g.node_len.resize(nn);
    std::memcpy(g.node_len.data(), p + off, nn * 8);
    off += nn * 8;
    uint64_t m = 0;
    if (!rd64(m)) return "truncated graph file " + path;
    std::map<std::string, std::vector<uint32_t>> paths;
    for (uint64_t i = 0; i < m; ++i) {
        uint64_t kl = 0, vl = 0;
        if (!rd64(kl) || kl > (1u << 20) || off + kl > n) return "corrupt graph file " + path;
        std::string key(p + off, kl);
        off += kl;
        if (!rd64(vl) || vl > (n - off) / 8) return "truncated graph file " + path;
        auto &dst = paths[key];
        dst.reserve(vl);
        for (uint64_t j = 0; j < vl; ++j) {
            uint64_t x;
            std::memcpy(&x, p + off + 8 * j, 8);
            if (x > 0xFFFFFFFFull) return "node index out of range in " + path;
            dst.push_back((uint32_t)x);
        }
        off += vl * 8;
    }
    finish_graph(paths, g);
    return "";
}

static std::string slurp(const std::string &path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return "cannot open serialized graph " + path;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    out.resize((size_t)n);
    if (n) f.read(&out[0], n);
    return f ? "" : "cannot read " + path;
}

std::string read_graph_bin(const std::string &path, HostGraph &g) {
    std::string buf;
    std::string e = slurp(path, buf);
    if (!e.empty()) return e;
    return parse_graph_bincode(buf.data(), buf.size(), path, g);
}

// ---- the two stream codecs of zip.rs:191-223 (`--lz`: LZ4 frame, lz4_flex; `--zstd`).  The image ships the shared
// libraries without headers, so the few entry points are bound at run time.
namespace {
struct ZstdIn { const void *src; size_t size, pos; };
struct ZstdOut { void *dst; size_t size, pos; };
void *codec_sym(const char *lib, const char *sym, std::string &err) {
    static std::map<std::string, void *> handles;
    void *&h = handles[lib];
    if (!h) h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("codec library ") + lib + " is not available on this host"; return nullptr; }
    void *f = dlsym(h, sym);
    if (!f) err = std::string(sym) + " missing from " + lib;
    return f;
}
std::string decode_lz4_frame(const std::string &in, std::string &out) {
    std::string err;
    auto create = (size_t(*)(void **, unsigned))codec_sym("liblz4.so.1", "LZ4F_createDecompressionContext", err);
    auto destroy = (size_t(*)(void *))codec_sym("liblz4.so.1", "LZ4F_freeDecompressionContext", err);
    auto dec = (size_t(*)(void *, void *, size_t *, const void *, size_t *, const void *))codec_sym("liblz4.so.1", "LZ4F_decompress", err);
    auto is_err = (unsigned (*)(size_t))codec_sym("liblz4.so.1", "LZ4F_isError", err);
    if (!create || !destroy || !dec || !is_err) return err;
    void *dctx = nullptr;
    if (is_err(create(&dctx, 100))) return "LZ4F_createDecompressionContext failed";
    std::vector<char> chunk(4u << 20);
    size_t pos = 0;
    out.clear();
    std::string res;
    while (pos < in.size()) {
        size_t dn = chunk.size(), sn = in.size() - pos;
        const size_t r = dec(dctx, chunk.data(), &dn, in.data() + pos, &sn, nullptr);
        if (is_err(r)) { res = "corrupt LZ4 frame"; break; }
        out.append(chunk.data(), dn);
        pos += sn;
        if (sn == 0 && dn == 0) { res = "truncated LZ4 frame"; break; }
    }
    destroy(dctx);
    return res;
}
std::string decode_zstd(const std::string &in, std::string &out) {
    std::string err;
    auto create = (void *(*)())codec_sym("libzstd.so.1", "ZSTD_createDStream", err);
    auto init = (size_t(*)(void *))codec_sym("libzstd.so.1", "ZSTD_initDStream", err);
    auto dec = (size_t(*)(void *, ZstdOut *, ZstdIn *))codec_sym("libzstd.so.1", "ZSTD_decompressStream", err);
    auto destroy = (size_t(*)(void *))codec_sym("libzstd.so.1", "ZSTD_freeDStream", err);
    auto is_err = (unsigned (*)(size_t))codec_sym("libzstd.so.1", "ZSTD_isError", err);
    if (!create || !init || !dec || !destroy || !is_err) return err;
    void *ds = create();
    if (!ds || is_err(init(ds))) return "ZSTD_createDStream failed";
    std::vector<char> chunk(4u << 20);
    ZstdIn zi{in.data(), in.size(), 0};
    out.clear();
    std::string res;
    size_t r = 1;
    while (zi.pos < zi.size || r != 0) {
        ZstdOut zo{chunk.data(), chunk.size(), 0};
        const size_t before = zi.pos;
        r = dec(ds, &zo, &zi);
        if (is_err(r)) { res = "corrupt zstd stream"; break; }
        out.append(chunk.data(), zo.pos);
        if (zi.pos == before && zo.pos == 0) { if (r != 0) res = "truncated zstd stream"; break; }
    }
    destroy(ds);
    return res;
}
}  // namespace

std::string read_graph_zip(const std::string &path, int codec, HostGraph &g) {
    std::string raw, img;
    std::string e = slurp(path, raw);
    if (!e.empty()) return e;
    e = codec == 2 ? decode_lz4_frame(raw, img) : decode_zstd(raw, img);
    if (!e.empty()) return e + " (" + path + ")";
    return parse_graph_bincode(img.data(), img.size(), path, g);
}

MappedFile::~MappedFile() {
    if (data && size) munmap(const_cast<char *>(data), size);
    if (fd >= 0) close(fd);
}
std::string MappedFile::open(const std::string &path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return "cannot open GAF mapping file " + path;
    struct stat st;
    if (fstat(fd, &st) != 0) return "cannot stat " + path;
    size = (size_t)st.st_size;
    if (size == 0) { data = nullptr; return ""; }
    void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) return "cannot mmap " + path;
    data = static_cast<const char *>(p);
    madvise(p, size, MADV_SEQUENTIAL);
    return "";
}

static inline uint64_t hash_bytes(const char *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 0x100000001b3ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    return h;
}
// parses an unsigned integer field; "*" or anything non-numeric => null
static inline bool parse_u32(const char *b, const char *e, uint32_t &out) {
    if (b == e) return false;
    uint64_t v = 0;
    for (const char *p = b; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (uint64_t)(*p - '0');
        if (v > 0xFFFFFFFFull) v = 0xFFFFFFFFull;
    }
    out = (uint32_t)v;
    return true;
}

static void parse_chunk(const char *b, const char *e, const char *base, HostReads &o) {
    const char *p = b;
    while (p < e) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(e - p)));
        const char *le = nl ? nl : e;
        const char *line_end = le;
        if (line_end > p && line_end[-1] == '\r') --line_end;
        if (line_end > p && *p != '@') {                       // comment prefix "@" (rcls.rs:123)
            const char *fb[12], *fe[12];
            int nf = 0;
            const char *q = p;
            while (nf < 12) {
                const char *t = static_cast<const char *>(memchr(q, '\t', (size_t)(line_end - q)));
                fb[nf] = q; fe[nf] = t ? t : line_end; ++nf;
                if (!t) break;
                q = t + 1;
            }
            uint8_t flag = 0;
            uint32_t ql = 0, ps = 0, pe = 0, pl = 0, mq = 255;
            if (nf > 1) parse_u32(fb[1], fe[1], ql);
            bool path_null = nf <= 5 || fe[5] == fb[5] || (fe[5] - fb[5] == 1 && *fb[5] == '*');   // '*' and the empty field are null (the reader's null_values / missing_is_null)
            if (!path_null) {
                for (const char *c = fb[5]; c < fe[5];) {       // regex \d+ over the walk (rcls.rs:242-245)
                    if (*c >= '0' && *c <= '9') {
                        uint64_t v = 0;
                        while (c < fe[5] && *c >= '0' && *c <= '9') { v = v * 10 + (uint64_t)(*c - '0'); if (v > 0xFFFFFFFFull) v = 0xFFFFFFFFull; ++c; }
                        o.node_id.push_back((uint32_t)v);
                    } else ++c;
                }
            } else flag |= 1;
            if (!(nf > 6 && parse_u32(fb[6], fe[6], pl))) flag |= 1;
            if (!(nf > 7 && parse_u32(fb[7], fe[7], ps))) flag |= 1;
            if (!(nf > 8 && parse_u32(fb[8], fe[8], pe))) flag |= 1;
            if (nf > 11) { uint32_t m; if (parse_u32(fb[11], fe[11], m)) mq = m > 255 ? 255 : m; }
            o.step_off.push_back((uint32_t)o.node_id.size());
            o.pstart.push_back(ps); o.pend.push_back(pe); o.qlen.push_back(ql);
            o.mapq.push_back((uint8_t)mq); o.flags.push_back(flag);
            o.id_hash.push_back(hash_bytes(fb[0], (size_t)(fe[0] - fb[0])));
            o.id_span.emplace_back((uint64_t)(fb[0] - base), (uint32_t)(fe[0] - fb[0]));
            ++o.n_lines;
        }
        p = nl ? nl + 1 : e;
    }
}

std::string parse_gaf(const MappedFile &mf, HostReads &out, int n_threads) {
    out = HostReads();
    if (mf.size == 0) return "";
    if (n_threads < 1) n_threads = 1;
    // split at line boundaries
    std::vector<const char *> cut{mf.data};
    for (int t = 1; t < n_threads; ++t) {
        const char *p = mf.data + mf.size * (size_t)t / (size_t)n_threads;
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(mf.data + mf.size - p)));
        const char *c = nl ? nl + 1 : mf.data + mf.size;
        if (c > cut.back()) cut.push_back(c);
    }
    cut.push_back(mf.data + mf.size);
    size_t nc = cut.size() - 1;
    std::vector<HostReads> parts(nc);
    std::vector<std::thread> th;
    for (size_t i = 0; i < nc; ++i) th.emplace_back([&, i] { parse_chunk(cut[i], cut[i + 1], mf.data, parts[i]); });
    for (auto &t : th) t.join();
    uint64_t T = 0, R = 0;
    for (auto &p : parts) { T += p.node_id.size(); R += p.pstart.size(); }
    if (T >= 0xFFFFFFFFull) return "GAF has more than 2^32 walk steps; split the input";
    out.step_off.reserve(R + 1); out.node_id.reserve(T);
    for (auto &p : parts) {
        uint32_t basev = (uint32_t)out.node_id.size();
        out.node_id.insert(out.node_id.end(), p.node_id.begin(), p.node_id.end());
        for (size_t i = 1; i < p.step_off.size(); ++i) out.step_off.push_back(basev + p.step_off[i]);
        out.pstart.insert(out.pstart.end(), p.pstart.begin(), p.pstart.end());
        out.pend.insert(out.pend.end(), p.pend.begin(), p.pend.end());
        out.qlen.insert(out.qlen.end(), p.qlen.begin(), p.qlen.end());
        out.mapq.insert(out.mapq.end(), p.mapq.begin(), p.mapq.end());
        out.flags.insert(out.flags.end(), p.flags.begin(), p.flags.end());
        out.id_hash.insert(out.id_hash.end(), p.id_hash.begin(), p.id_hash.end());
        out.id_span.insert(out.id_span.end(), p.id_span.begin(), p.id_span.end());
        out.n_lines += p.n_lines;
        p = HostReads();
    }
    return "";
}

std::string fmt_f64(double v) {
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);   // shortest round-trip
    std::string s(buf, r.ptr);
    size_t e = s.find('e');
    if (e != std::string::npos) {                          // "1e-05" -> "1e-5" (ryu style)
        std::string mant = s.substr(0, e), ex = s.substr(e + 1);
        bool neg = !ex.empty() && ex[0] == '-';
        if (!ex.empty() && (ex[0] == '-' || ex[0] == '+')) ex = ex.substr(1);
        while (ex.size() > 1 && ex[0] == '0') ex = ex.substr(1);
        return mant + "e" + (neg ? "-" : "") + ex;
    }
    if (s.find('.') == std::string::npos && s.find("inf") == std::string::npos && s.find("nan") == std::string::npos) s += ".0";
    return s;
}

}  // namespace ptx
