// oracle/blob.h -- TEST INFRASTRUCTURE.  Tiny named-array container used for fixtures
// (tests/golden/*.bin).  Layout: "PVB1", u32 n_entries, then per entry
//   char name[48] (NUL padded), u32 dtype (0=f32 1=u32 2=i32 3=u64), u64 count, payload.
// The Python side is cs348b-pbrt_amd/blob.py.
#ifndef ORC_BLOB_H
#define ORC_BLOB_H
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>

namespace blob {
enum { F32 = 0, U32 = 1, I32 = 2, U64 = 3 };
struct Entry {
    uint32_t dtype;
    std::vector<unsigned char> bytes;
    size_t count() const { return bytes.size() / (dtype == U64 ? 8 : 4); }
    const float *f32() const { return (const float *)bytes.data(); }
    const uint32_t *u32() const { return (const uint32_t *)bytes.data(); }
    const int32_t *i32() const { return (const int32_t *)bytes.data(); }
    const uint64_t *u64() const { return (const uint64_t *)bytes.data(); }
};
struct Blob {
    std::vector<std::string> order;
    std::map<std::string, Entry> e;
    void put(const std::string &name, uint32_t dtype, const void *data, size_t count) {
        if (!e.count(name)) order.push_back(name);
        Entry &en = e[name];
        en.dtype = dtype;
        size_t nb = count * (dtype == U64 ? 8 : 4);
        en.bytes.assign((const unsigned char *)data, (const unsigned char *)data + nb);
    }
    void putf(const std::string &name, const float *d, size_t n) { put(name, F32, d, n); }
    void putf(const std::string &name, const std::vector<float> &v) { put(name, F32, v.data(), v.size()); }
    void putf1(const std::string &name, float v) { put(name, F32, &v, 1); }
    void putu(const std::string &name, const std::vector<uint32_t> &v) { put(name, U32, v.data(), v.size()); }
    void putu1(const std::string &name, uint32_t v) { put(name, U32, &v, 1); }
    void puti1(const std::string &name, int32_t v) { put(name, I32, &v, 1); }
    bool has(const std::string &n) const { return e.count(n) != 0; }
    const Entry &get(const std::string &n) const {
        std::map<std::string, Entry>::const_iterator it = e.find(n);
        if (it == e.end()) { fprintf(stderr, "blob: missing entry '%s'\n", n.c_str()); abort(); }
        return it->second;
    }
    bool save(const char *path) const {
        FILE *f = fopen(path, "wb");
        if (!f) return false;
        fwrite("PVB1", 1, 4, f);
        uint32_t n = (uint32_t)order.size();
        fwrite(&n, 4, 1, f);
        for (size_t i = 0; i < order.size(); ++i) {
            const Entry &en = e.find(order[i])->second;
            char name[48];
            memset(name, 0, sizeof(name));
            strncpy(name, order[i].c_str(), 47);
            fwrite(name, 1, 48, f);
            fwrite(&en.dtype, 4, 1, f);
            uint64_t c = en.count();
            fwrite(&c, 8, 1, f);
            fwrite(en.bytes.data(), 1, en.bytes.size(), f);
        }
        fclose(f);
        return true;
    }
    bool load(const char *path) {
        FILE *f = fopen(path, "rb");
        if (!f) return false;
        char magic[4];
        uint32_t n;
        if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "PVB1", 4) || fread(&n, 4, 1, f) != 1) { fclose(f); return false; }
        for (uint32_t i = 0; i < n; ++i) {
            char name[48];
            uint32_t dtype;
            uint64_t c;
            if (fread(name, 1, 48, f) != 48 || fread(&dtype, 4, 1, f) != 1 || fread(&c, 8, 1, f) != 1) { fclose(f); return false; }
            name[47] = 0;
            Entry en;
            en.dtype = dtype;
            en.bytes.resize(c * (dtype == U64 ? 8 : 4));
            if (c && fread(en.bytes.data(), 1, en.bytes.size(), f) != en.bytes.size()) { fclose(f); return false; }
            order.push_back(name);
            e[name] = en;
        }
        fclose(f);
        return true;
    }
};
}  // namespace blob
#endif
