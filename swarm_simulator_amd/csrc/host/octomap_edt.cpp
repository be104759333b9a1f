// Octomap ".bt" reader and the distance grid the corridor stage samples.
//
// The reference gets both from system libraries that are not vendored:
//   new octomap::OcTree(path)                                   swarm_traj_planner_rbp_test_all.cpp:51
//   DynamicEDTOctomap(maxDist=1, tree, min, max, false).update() swarm_traj_planner_rbp_test_all.cpp:57-63
// This file restates their published formats/semantics (octomap 1.9 binary tree stream; dynamicEDT3D's
// clamped Euclidean distance in cells) — SURVEY.md Appendix B.  The EDT here is the exact one
// (Felzenszwalb/Huttenlocher lower-envelope passes); dynamicEDT3D's brushfire agrees with it at the
// short ranges the planner thresholds at (< 4 cells).
#include "rbp_host.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct BtReader {
    const unsigned char* p;
    const unsigned char* end;
    std::vector<int32_t>* leaves;  // (kx,ky,kz,size) of occupied leaves, keys relative to the tree centre
    int64_t nodes = 0;
    bool ok = true;
    // node = 2 bytes, 2 bits per child: (bit 2i, bit 2i+1) = (1,0) free leaf, (0,1) occupied leaf, (1,1) inner
    void node(int depth, int kx, int ky, int kz) {
        ++nodes;
        if (end - p < 2) {
            ok = false;
            return;
        }
        unsigned bits = p[0] | (p[1] << 8);
        p += 2;
        int half = 1 << (15 - depth);  // child edge length in voxels
        int kind[8];
        for (int i = 0; i < 8; ++i) kind[i] = (bits >> (2 * i)) & 3;
        for (int i = 0; i < 8 && ok; ++i) {
            int cx = kx + ((i & 1) ? half : 0), cy = ky + ((i & 2) ? half : 0), cz = kz + ((i & 4) ? half : 0);
            if (kind[i] == 2) {  // bit0 = 0, bit1 = 1 : occupied leaf
                ++nodes;
                leaves->push_back(cx);
                leaves->push_back(cy);
                leaves->push_back(cz);
                leaves->push_back(half);
            } else if (kind[i] == 1) {  // free leaf
                ++nodes;
            } else if (kind[i] == 3) {
                if (depth + 1 >= 16) {
                    ok = false;
                    return;
                }
                node(depth + 1, cx, cy, cz);
            }
        }
    }
};

// 1-D squared distance transform (lower envelope of parabolas), f in/out, n <= 4096
void dt1d(const double* f, double* d, int n, int* v, double* z) {
    const double INF = 1e20;
    int k = 0;
    v[0] = 0;
    z[0] = -INF;
    z[1] = INF;
    for (int q = 1; q < n; ++q) {
        if (f[q] >= INF) continue;
        if (f[v[k]] >= INF) {  // first finite site
            v[k] = q;
            continue;
        }
        double s;
        while (true) {
            s = ((f[q] + (double)q * q) - (f[v[k]] + (double)v[k] * v[k])) / (2.0 * q - 2.0 * v[k]);
            if (s <= z[k] && k > 0)
                --k;
            else
                break;
        }
        ++k;
        v[k] = q;
        z[k] = s;
        z[k + 1] = INF;
    }
    k = 0;
    for (int q = 0; q < n; ++q) {
        while (z[k + 1] < q) ++k;
        double dq = (double)(q - v[k]);
        d[q] = (f[v[k]] >= INF) ? INF : dq * dq + f[v[k]];
    }
}

}  // namespace

extern "C" int rbp_octomap_load_bt(const char* path, rbp_octomap_buf* out) {
    if (!path || !out) return RBP_ERR_BAD_ARGUMENT;
    memset(out, 0, sizeof(*out));
    FILE* f = fopen(path, "rb");
    if (!f) return RBP_ERR_BAD_ARGUMENT;
    std::string data;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, n);
    fclose(f);
    // text header: "# Octomap OcTree binary file", comment lines, "id OcTree", "size N", "res R", "data"
    size_t pos = 0;
    double res = 0;
    long long size = -1;
    bool have_data = false, id_ok = false;
    while (pos < data.size()) {
        size_t eol = data.find('\n', pos);
        if (eol == std::string::npos) break;
        std::string line = data.substr(pos, eol - pos);
        pos = eol + 1;
        if (line.empty() || line[0] == '#') continue;
        if (line.rfind("id ", 0) == 0) id_ok = line.find("OcTree") != std::string::npos;
        else if (line.rfind("size ", 0) == 0) size = atoll(line.c_str() + 5);
        else if (line.rfind("res ", 0) == 0) res = atof(line.c_str() + 4);
        else if (line.rfind("data", 0) == 0) {
            have_data = true;
            break;
        }
    }
    if (!have_data || !id_ok || res <= 0) return RBP_ERR_BAD_ARGUMENT;
    std::vector<int32_t> leaves;
    BtReader rd{(const unsigned char*)data.data() + pos, (const unsigned char*)data.data() + data.size(), &leaves};
    if (size > 0) rd.node(0, -32768, -32768, -32768);
    if (!rd.ok) return RBP_ERR_BAD_ARGUMENT;
    out->res = res;
    out->n_occupied = (int64_t)(leaves.size() / 4);
    out->n_nodes = rd.nodes;
    out->keys = (int32_t*)malloc(sizeof(int32_t) * (leaves.size() ? leaves.size() : 4));
    memcpy(out->keys, leaves.data(), sizeof(int32_t) * leaves.size());
    return RBP_OK;
}

extern "C" void rbp_octomap_free(rbp_octomap_buf* m) {
    if (!m) return;
    free(m->keys);
    memset(m, 0, sizeof(*m));
}

extern "C" int rbp_world_build(const rbp_octomap_buf* map, const double bbx_min[3], const double bbx_max[3],
                               double max_dist, rbp_world_buf* out) {
    if (!map || !out || !bbx_min || !bbx_max) return RBP_ERR_BAD_ARGUMENT;
    memset(out, 0, sizeof(*out));
    const double res = map->res;
    const double rf = 1.0 / res;  // octomap resolution_factor
    int kmin[3], kmax[3];
    for (int a = 0; a < 3; ++a) {
        // octomap::point3d is float32; coordToKey(c) = (int)floor(resolution_factor * c) (+32768)
        kmin[a] = (int)floor(rf * (double)(float)bbx_min[a]);
        kmax[a] = (int)floor(rf * (double)(float)bbx_max[a]);
        if (kmax[a] < kmin[a]) return RBP_ERR_BAD_ARGUMENT;
        out->key_min[a] = kmin[a];
        out->dim[a] = kmax[a] - kmin[a] + 1;
        if (out->dim[a] > 4096) return RBP_ERR_BAD_ARGUMENT;
    }
    out->res = res;
    const int nx = out->dim[0], ny = out->dim[1], nz = out->dim[2];
    const size_t ncell = (size_t)nx * ny * nz;
    const double INF = 1e20;
    std::vector<double> g(ncell, INF);
    // occupied leaves, clipped to the bounding box (DynamicEDTOctomap::initializeOcTree)
    for (int64_t i = 0; i < map->n_occupied; ++i) {
        const int32_t* k = map->keys + 4 * i;
        int s = k[3];
        int x0 = std::max(k[0], kmin[0]), x1 = std::min(k[0] + s - 1, kmax[0]);
        int y0 = std::max(k[1], kmin[1]), y1 = std::min(k[1] + s - 1, kmax[1]);
        int z0 = std::max(k[2], kmin[2]), z1 = std::min(k[2] + s - 1, kmax[2]);
        for (int x = x0; x <= x1; ++x)
            for (int y = y0; y <= y1; ++y)
                for (int z = z0; z <= z1; ++z) g[((size_t)(x - kmin[0]) * ny + (y - kmin[1])) * nz + (z - kmin[2])] = 0.0;
    }
    // separable exact squared EDT
    int nmax = std::max(nx, std::max(ny, nz));
    std::vector<double> f(nmax), d(nmax), zb(nmax + 1);
    std::vector<int> v(nmax);
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            double* row = &g[((size_t)x * ny + y) * nz];
            for (int z = 0; z < nz; ++z) f[z] = row[z];
            dt1d(f.data(), d.data(), nz, v.data(), zb.data());
            for (int z = 0; z < nz; ++z) row[z] = d[z];
        }
    for (int x = 0; x < nx; ++x)
        for (int z = 0; z < nz; ++z) {
            for (int y = 0; y < ny; ++y) f[y] = g[((size_t)x * ny + y) * nz + z];
            dt1d(f.data(), d.data(), ny, v.data(), zb.data());
            for (int y = 0; y < ny; ++y) g[((size_t)x * ny + y) * nz + z] = d[y];
        }
    for (int y = 0; y < ny; ++y)
        for (int z = 0; z < nz; ++z) {
            for (int x = 0; x < nx; ++x) f[x] = g[((size_t)x * ny + y) * nz + z];
            dt1d(f.data(), d.data(), nx, v.data(), zb.data());
            for (int x = 0; x < nx; ++x) g[((size_t)x * ny + y) * nz + z] = d[x];
        }
    // dynamicEDT3D: maxDist_squared = ((int)(maxDist/res + 1))^2, cells at or beyond keep dist = sqrt(maxDist_squared)
    const int md = (int)(max_dist / res + 1);
    const long long md2 = (long long)md * md;
    out->dist = (float*)malloc(sizeof(float) * ncell);
    for (size_t i = 0; i < ncell; ++i) {
        double sq = g[i];
        float cells = (sq < (double)md2) ? (float)std::sqrt(sq) : (float)std::sqrt((double)md2);
        out->dist[i] = (float)((double)cells * res);  // getDistance: float dist * double treeResolution -> float
    }
    return RBP_OK;
}

extern "C" void rbp_world_free(rbp_world_buf* w) {
    if (!w) return;
    free(w->dist);
    memset(w, 0, sizeof(*w));
}
