// lsf_host_stl.hpp -- lsf_stl_read / lsf_stl_get: the reference's binary STL ingest (subs.f90:17-121) with a hash instead of its quadratic
// vertex search.  Host code only.  Included by lsf_api.hip inside extern "C".
#pragma once

// ---- binary STL with the reference's vertex merge (subs.f90:17-121) --------------------------------------------
namespace {
struct StlResult {
    std::vector<float> nodes;   // 3 per node
    std::vector<int32_t> elem;  // 3 per triangle, 1-based
};
thread_local StlResult g_stl;
} // namespace

int lsf_stl_read(const char* path, int* nSurfElem, int* nSurfNode)
{
    Trace trace_("lsf_stl_read");
    if (!path || !nSurfElem || !nSurfNode) return fail(LSF_ERR_INVALID, "NULL pointer");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(LSF_ERR_INVALID, std::string("cannot open ") + path);
    unsigned char head[84];
    if (fread(head, 1, 84, f) != 84) {
        fclose(f);
        return fail(LSF_ERR_INVALID, "STL file shorter than its header");
    }
    int32_t ntri = 0;
    std::memcpy(&ntri, head + 80, 4); // subs.f90:38-39
    if (ntri < 1) {
        fclose(f);
        return fail(LSF_ERR_INVALID, "STL file holds no triangle");
    }
    std::vector<unsigned char> rec((size_t)ntri * 50); // normal, 3 vertices (REAL*4), INTEGER*2 padding: subs.f90:47-53
    const size_t got = fread(rec.data(), 1, rec.size(), f);
    fclose(f);
    if (got != rec.size()) return fail(LSF_ERR_INVALID, "STL file shorter than its triangle count");
    StlResult& R = g_stl;
    R.nodes.clear();
    R.elem.assign((size_t)ntri * 3, 0);
    // Merge (subs.f90:64-93).  Two REAL*4 values can differ by less than 1e-13 without being equal only below 2^-19
    // (above it neighbouring floats are >= 1.1e-13 apart, also across that threshold), so a coordinate is keyed by its
    // bits when it is large and by one shared key when it is small; candidates of a key are kept in node order and tested
    // with the reference's own predicate, the first one inside the search bound wins.
    struct Key {
        uint32_t a, b, c;
        bool operator==(const Key& o) const { return a == o.a && b == o.b && c == o.c; }
    };
    struct KeyHash {
        size_t operator()(const Key& k) const { return ((size_t)k.a * 0x9E3779B1u) ^ ((size_t)k.b * 0x85EBCA77u << 1) ^ ((size_t)k.c * 0xC2B2AE3Du << 2); }
    };
    auto key1 = [](float v) -> uint32_t {
        if (std::fabs(v) < 1.9073486328125e-06f) return 0xFFFFFFFFu; // 2^-19
        uint32_t u;
        std::memcpy(&u, &v, 4);
        return u == 0x80000000u ? 0u : u;
    };
    std::unordered_map<Key, std::vector<int32_t>, KeyHash> map;
    map.reserve((size_t)ntri);
    int32_t bound = 3, k = 0; // nSurfNode (search bound) and the number of nodes so far
    for (int32_t n = 0; n < ntri; ++n) {
        for (int p = 0; p < 3; ++p) {
            float v[3];
            std::memcpy(v, rec.data() + (size_t)n * 50 + 12 + 12 * p, 12);
            const Key key{key1(v[0]), key1(v[1]), key1(v[2])};
            int32_t share = 0;
            auto it = map.find(key);
            if (it != map.end())
                for (int32_t cand : it->second) { // ascending node numbers
                    if (cand > bound) break;
                    const float* q = &R.nodes[(size_t)(cand - 1) * 3];
                    if ((double)std::fabs(q[0] - v[0]) < 1.e-13 && (double)std::fabs(q[1] - v[1]) < 1.e-13 &&
                        (double)std::fabs(q[2] - v[2]) < 1.e-13) {
                        share = cand;
                        break;
                    }
                }
            if (share > 0) {
                R.elem[(size_t)n * 3 + p] = share;
            } else {
                ++k;
                R.nodes.insert(R.nodes.end(), v, v + 3);
                R.elem[(size_t)n * 3 + p] = k;
                map[key].push_back(k);
            }
        }
        bound = k; // subs.f90:91
    }
    *nSurfElem = ntri;
    *nSurfNode = k;
    return LSF_OK;
}

int lsf_stl_get(double* surfX, int32_t* surfElem)
{
    if (!surfX || !surfElem) return fail(LSF_ERR_INVALID, "NULL pointer");
    StlResult& R = g_stl;
    if (R.elem.empty()) return fail(LSF_ERR_INVALID, "lsf_stl_get without lsf_stl_read");
    const size_t nn = R.nodes.size() / 3, nt = R.elem.size() / 3;
    for (size_t q = 0; q < nn; ++q)
        for (int c = 0; c < 3; ++c) surfX[q + nn * c] = (double)R.nodes[q * 3 + c]; // REAL*4 -> REAL(8), subs.f90:99-103
    for (size_t t = 0; t < nt; ++t)
        for (int p = 0; p < 3; ++p) surfElem[t + nt * p] = R.elem[t * 3 + p];
    R = StlResult{};
    return LSF_OK;
}
