// ECBS front-end: discrete initial trajectories (initTraj) and segment times T.
//
// Host-side, sequential, integer search — it FEEDS the hot path and is not part of it.
// Behavioural model: ECBSPlanner (swarm_planner/include/ecbs_planner.hpp:21-136) on top of the vendored
// libMultiRobotPlanning ECBS with the reference's 3-D, size-aware conflict rules
// (third_party/ecbs/include/environment.hpp:467-524 neighbours, :656-681 conflicts, :526-610 first
// conflict / constraints; ecbs.hpp:109-297 high level; a_star_epsilon.hpp:86-285 low level).
// Own implementation on std::set queues with explicit, deterministic tie-breaks (node id); the reference's
// ties are resolved by Boost.Heap internals, so bit-identical paths are not a goal:
// the output is judged on validity (conflict-free under the same rules) and bounded sub-optimality.
#include "rbp_host.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Cell {
    int x, y, z;
    bool operator==(const Cell& o) const { return x == o.x && y == o.y && z == o.z; }
};

inline uint64_t pack4(int t, int x, int y, int z) {
    return ((uint64_t)(uint16_t)t << 48) | ((uint64_t)(uint16_t)x << 32) | ((uint64_t)(uint16_t)y << 16) | (uint16_t)z;
}

struct Constraints {
    std::unordered_set<uint64_t> vertex;                       // (t,x,y,z)
    std::set<std::tuple<int, int, int, int, int, int, int>> edge;  // (t, from, to)
};

struct Grid {
    int dimx, dimy, dimz;
    std::vector<uint8_t> obstacle;
    double grid_size;  // grid_xy_res only (ecbs_planner.hpp:22-23)
    std::vector<double> radius;
    bool blocked(int x, int y, int z) const {
        return x < 0 || y < 0 || z < 0 || x >= dimx || y >= dimy || z >= dimz || obstacle[((size_t)x * dimy + y) * dimz + z];
    }
};

using Path = std::vector<Cell>;  // state at time t = path[min(t, size-1)]

inline const Cell& at(const Path& p, int t) { return p[(size_t)t < p.size() ? t : p.size() - 1]; }

// environment.hpp:656-664
bool vertex_conflict(const Grid& g, int i, int j, const Cell& a, const Cell& b) {
    double rr = g.radius[i] + g.radius[j];
    if (rr < g.grid_size) return a == b;
    double dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
    return std::sqrt(dx * dx + dy * dy + dz * dz) * g.grid_size < rr;
}

// environment.hpp:69-93 (closest approach of the relative motion to the origin) and :666-681
bool edge_conflict(const Grid& g, int i, int j, const Cell& a1, const Cell& b1, const Cell& a2, const Cell& b2) {
    double rr = g.radius[i] + g.radius[j];
    if (rr < g.grid_size * 0.5) return a1 == b2 && b1 == a2;
    double ax = a2.x - a1.x, ay = a2.y - a1.y, az = a2.z - a1.z;
    double bx = b2.x - b1.x, by = b2.y - b1.y, bz = b2.z - b1.z;
    double md = std::sqrt(ax * ax + ay * ay + az * az);
    if (!(ax == bx && ay == by && az == bz)) {
        double d = std::sqrt(bx * bx + by * by + bz * bz);
        if (md > d) md = d;
        double nx = bx - ax, ny = by - ay, nz = bz - az;
        double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
        nx /= nn, ny /= nn, nz /= nn;
        double adn = ax * nx + ay * ny + az * nz;
        double cx = ax - nx * adn, cy = ay - ny * adn, cz = az - nz * adn;
        d = std::sqrt(cx * cx + cy * cy + cz * cz);
        if ((cx - ax) * (cx - bx) + (cy - ay) * (cy - by) + (cz - az) * (cz - bz) < 0 && md > d) md = d;
    }
    return md * g.grid_size <= rr;
}

int count_conflicts(const Grid& g, const std::vector<Path>& sol) {  // environment.hpp:425-460
    int max_t = 0;
    for (auto& p : sol) max_t = std::max<int>(max_t, (int)p.size() - 1);
    int n = 0;
    const int N = (int)sol.size();
    for (int t = 0; t < max_t; ++t) {
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j)
                if (vertex_conflict(g, i, j, at(sol[i], t), at(sol[j], t))) ++n;
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j)
                if (edge_conflict(g, i, j, at(sol[i], t), at(sol[i], t + 1), at(sol[j], t), at(sol[j], t + 1))) ++n;
    }
    return n;
}

struct Conflict {
    bool is_edge;
    int t, i, j;
    Cell a1, b1, a2, b2;
};

bool first_conflict(const Grid& g, const std::vector<Path>& sol, Conflict& c) {  // environment.hpp:526-589
    int max_t = 0;
    for (auto& p : sol) max_t = std::max<int>(max_t, (int)p.size() - 1);
    const int N = (int)sol.size();
    for (int t = 0; t < max_t; ++t) {
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j)
                if (vertex_conflict(g, i, j, at(sol[i], t), at(sol[j], t))) {
                    c = {false, t, i, j, at(sol[i], t), at(sol[i], t), at(sol[j], t), at(sol[j], t)};
                    return true;
                }
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j)
                if (edge_conflict(g, i, j, at(sol[i], t), at(sol[i], t + 1), at(sol[j], t), at(sol[j], t + 1))) {
                    c = {true, t, i, j, at(sol[i], t), at(sol[i], t + 1), at(sol[j], t), at(sol[j], t + 1)};
                    return true;
                }
    }
    return false;
}

// ---- low level: A*-epsilon over (t,x,y,z) ------------------------------------------------------
struct LLNode {
    int t, x, y, z;
    int g, f, focal;
    int parent;
};

struct LowLevelResult {
    Path path;
    int cost = 0, fmin = 0;
};

bool low_level(const Grid& g, int agent, const Cell& start, const Cell& goal, const Constraints& cons,
               const std::vector<Path>& others, float w, int64_t& expanded, LowLevelResult& out) {
    int last_goal_constraint = -1;  // environment.hpp:373-383
    for (uint64_t v : cons.vertex) {
        int t = (int)(v >> 48), x = (int)((v >> 32) & 0xffff), y = (int)((v >> 16) & 0xffff), z = (int)(v & 0xffff);
        if (x == goal.x && y == goal.y && z == goal.z) last_goal_constraint = std::max(last_goal_constraint, t);
    }
    const int t_limit = 8 * (g.dimx + g.dimy + g.dimz) + 64 + last_goal_constraint;
    auto heur = [&](int x, int y, int z) { return std::abs(x - goal.x) + std::abs(y - goal.y) + std::abs(z - goal.z); };
    auto focal_state = [&](int t, const Cell& c) {  // environment.hpp:392-405
        int n = 0;
        for (size_t i = 0; i < others.size(); ++i)
            if ((int)i != agent && !others[i].empty() && vertex_conflict(g, agent, (int)i, c, at(others[i], t))) ++n;
        return n;
    };
    auto focal_trans = [&](int t, const Cell& a, const Cell& b) {  // environment.hpp:408-422
        int n = 0;
        for (size_t i = 0; i < others.size(); ++i)
            if ((int)i != agent && !others[i].empty() &&
                edge_conflict(g, agent, (int)i, a, b, at(others[i], t), at(others[i], t + 1)))
                ++n;
        return n;
    };

    std::vector<LLNode> nodes;
    nodes.reserve(4096);
    using OpenKey = std::tuple<int, int, int>;        // f asc, g desc (-g), id
    using FocalKey = std::tuple<int, int, int, int>;  // focal, f, -g, id
    std::set<OpenKey> open;
    std::set<FocalKey> focal;
    std::unordered_map<uint64_t, int> best;  // state -> node id currently in open
    std::unordered_set<uint64_t> closed;

    nodes.push_back({0, start.x, start.y, start.z, 0, heur(start.x, start.y, start.z), 0, -1});
    open.insert({nodes[0].f, 0, 0});
    focal.insert({0, nodes[0].f, 0, 0});
    best[pack4(0, start.x, start.y, start.z)] = 0;
    int best_f = nodes[0].f;

    static const int DX[7] = {0, -1, 1, 0, 0, 0, 0};  // Wait, Left, Right, Up(y+1), Down, Top(z+1), Bottom
    static const int DY[7] = {0, 0, 0, 1, -1, 0, 0};
    static const int DZ[7] = {0, 0, 0, 0, 0, 1, -1};

    while (!open.empty()) {
        int old_best = best_f;
        best_f = std::get<0>(*open.begin());
        if (best_f > old_best) {  // a_star_epsilon.hpp:134-153: widen focal to the new bound
            for (auto it = open.begin(); it != open.end(); ++it) {
                int f = std::get<0>(*it);
                if (f > best_f * w) break;
                if (f > old_best * w) {
                    int id = std::get<2>(*it);
                    focal.insert({nodes[id].focal, nodes[id].f, -nodes[id].g, id});
                }
            }
        }
        int cur = std::get<3>(*focal.begin());
        LLNode n = nodes[cur];
        ++expanded;
        if (n.x == goal.x && n.y == goal.y && n.z == goal.z && n.t > last_goal_constraint) {
            out.path.clear();
            for (int id = cur; id >= 0; id = nodes[id].parent) out.path.push_back({nodes[id].x, nodes[id].y, nodes[id].z});
            std::reverse(out.path.begin(), out.path.end());
            out.cost = n.g;
            out.fmin = std::get<0>(*open.begin());
            return true;
        }
        focal.erase(focal.begin());
        open.erase({n.f, -n.g, cur});
        uint64_t key = pack4(n.t, n.x, n.y, n.z);
        best.erase(key);
        closed.insert(key);
        if (n.t >= t_limit) continue;
        for (int a = 0; a < 7; ++a) {
            int x = n.x + DX[a], y = n.y + DY[a], z = n.z + DZ[a], t = n.t + 1;
            if (g.blocked(x, y, z)) continue;
            uint64_t k2 = pack4(t, x, y, z);
            if (cons.vertex.count(k2)) continue;
            if (!cons.edge.empty() && cons.edge.count({n.t, n.x, n.y, n.z, x, y, z})) continue;
            if (closed.count(k2)) continue;
            int g2 = n.g + 1;
            int f2 = g2 + heur(x, y, z);
            int foc = n.focal + focal_state(t, {x, y, z}) + focal_trans(n.t, {n.x, n.y, n.z}, {x, y, z});
            auto it = best.find(k2);
            if (it == best.end()) {
                int id = (int)nodes.size();
                nodes.push_back({t, x, y, z, g2, f2, foc, cur});
                open.insert({f2, -g2, id});
                best[k2] = id;
                if (f2 <= best_f * w) focal.insert({foc, f2, -g2, id});
            } else {
                // unit costs: a state at time t always has g == t, so no cheaper re-discovery exists
                // (a_star_epsilon.hpp:246-270 would decrease-key here); keep the first one found.
            }
        }
    }
    return false;
}

struct HLNode {
    std::vector<Path> sol;
    std::vector<int> cost, fmin;
    std::vector<std::shared_ptr<Constraints>> cons;
    int total = 0, lb = 0, focal = 0, id = 0;
};

float world_distance(const rbp_world_buf* w, float x, float y, float z) {
    const double rf = 1.0 / w->res;
    int kx = (int)std::floor(rf * (double)x) - w->key_min[0];
    int ky = (int)std::floor(rf * (double)y) - w->key_min[1];
    int kz = (int)std::floor(rf * (double)z) - w->key_min[2];
    if (kx < 0 || ky < 0 || kz < 0 || kx >= w->dim[0] || ky >= w->dim[1] || kz >= w->dim[2]) return -1.0f;
    return w->dist[((size_t)kx * w->dim[1] + ky) * w->dim[2] + kz];
}

}  // namespace

extern "C" int rbp_ecbs_plan(const rbp_world_buf* world, const rbp_mission* mission, const rbp_param* param,
                             int64_t max_high_level_nodes, rbp_init_traj_buf* out) {
    if (!world || !mission || !param || !out) return RBP_ERR_BAD_ARGUMENT;
    memset(out, 0, sizeof(*out));
    const double eps = 1e-9;  // SP_EPSILON
    const int N = mission->N;
    // init_traj_planner.hpp:19-29
    double gmin[3], gmax[3], gres[3] = {param->grid_xy_res, param->grid_xy_res, param->grid_z_res};
    int dim[3];
    for (int a = 0; a < 3; ++a) {
        gmin[a] = std::ceil((param->world_min[a] - eps) / gres[a]) * gres[a];
        gmax[a] = std::floor((param->world_max[a] + eps) / gres[a]) * gres[a];
        dim[a] = (int)std::round((gmax[a] - gmin[a]) / gres[a]) + 1;
        if (dim[a] <= 0) return RBP_ERR_BAD_ARGUMENT;
    }
    Grid g;
    g.dimx = dim[0], g.dimy = dim[1], g.dimz = dim[2];
    g.grid_size = param->grid_xy_res;
    g.radius.assign(mission->radius, mission->radius + N);
    g.obstacle.assign((size_t)dim[0] * dim[1] * dim[2], 0);
    // ecbs_planner.hpp:80-109
    double r = 0;
    for (int qi = 0; qi < N; ++qi) r = std::max(r, mission->radius[qi]);
    for (double k = gmin[2]; k < gmax[2] + eps; k += gres[2])
        for (double i = gmin[0]; i < gmax[0] + eps; i += gres[0])
            for (double j = gmin[1]; j < gmax[1] + eps; j += gres[1]) {
                float d = world_distance(world, (float)i, (float)j, (float)k);
                if (d < 0) return 1;
                if (d < r + param->grid_margin) {
                    int x = (int)std::round((i - gmin[0]) / gres[0]);
                    int y = (int)std::round((j - gmin[1]) / gres[1]);
                    int z = (int)std::round((k - gmin[2]) / gres[2]);
                    if (x >= 0 && y >= 0 && z >= 0 && x < dim[0] && y < dim[1] && z < dim[2])
                        g.obstacle[((size_t)x * dim[1] + y) * dim[2] + z] = 1;
                }
            }
    // ecbs_planner.hpp:112-136
    std::vector<Cell> starts(N), goals(N);
    for (int i = 0; i < N; ++i) {
        for (int a = 0; a < 3; ++a) {
            int s = (int)std::round((mission->start[9 * i + a] - gmin[a]) / gres[a]);
            int e = (int)std::round((mission->goal[9 * i + a] - gmin[a]) / gres[a]);
            (a == 0 ? starts[i].x : a == 1 ? starts[i].y : starts[i].z) = s;
            (a == 0 ? goals[i].x : a == 1 ? goals[i].y : goals[i].z) = e;
        }
        if (g.blocked(starts[i].x, starts[i].y, starts[i].z) || g.blocked(goals[i].x, goals[i].y, goals[i].z)) return 1;
    }

    const float w = (float)param->ecbs_w;  // stored as float, ecbs.hpp:107
    int64_t ll_expanded = 0, hl_expanded = 0;

    auto root = std::make_shared<HLNode>();
    root->sol.resize(N);
    root->cost.assign(N, 0);
    root->fmin.assign(N, 0);
    root->cons.resize(N);
    for (int i = 0; i < N; ++i) root->cons[i] = std::make_shared<Constraints>();
    for (int i = 0; i < N; ++i) {
        LowLevelResult res;
        if (!low_level(g, i, starts[i], goals[i], *root->cons[i], root->sol, w, ll_expanded, res)) return 2;
        root->sol[i] = res.path;
        root->cost[i] = res.cost;
        root->fmin[i] = res.fmin;
        root->total += res.cost;
        root->lb += res.fmin;
    }
    root->focal = count_conflicts(g, root->sol);

    std::vector<std::shared_ptr<HLNode>> all{root};
    std::set<std::tuple<int, int>> open;         // (cost, id)
    std::set<std::tuple<int, int, int>> focal;   // (conflicts, cost, id)
    open.insert({root->total, 0});
    focal.insert({root->focal, root->total, 0});
    int best_cost = root->total;
    std::shared_ptr<HLNode> goal_node;
    int next_id = 1;
    while (!open.empty()) {
        int old_best = best_cost;
        best_cost = std::get<0>(*open.begin());
        if (best_cost > old_best) {  // ecbs.hpp:171-191
            for (auto& o : open) {
                int c = std::get<0>(o);
                if (c > best_cost * w) break;
                if (c > old_best * w) {
                    auto& n = all[std::get<1>(o)];
                    focal.insert({n->focal, n->total, n->id});
                }
            }
        }
        int id = std::get<2>(*focal.begin());
        std::shared_ptr<HLNode> P = all[id];
        focal.erase(focal.begin());
        open.erase({P->total, id});
        ++hl_expanded;
        Conflict c;
        if (!first_conflict(g, P->sol, c)) {
            goal_node = P;
            break;
        }
        if (max_high_level_nodes > 0 && hl_expanded > max_high_level_nodes) return 2;
        for (int side = 0; side < 2; ++side) {
            int ag = side == 0 ? c.i : c.j;
            auto child = std::make_shared<HLNode>(*P);
            child->id = next_id++;
            auto nc = std::make_shared<Constraints>(*P->cons[ag]);
            const Cell& a = side == 0 ? c.a1 : c.a2;
            const Cell& b = side == 0 ? c.b1 : c.b2;
            if (c.is_edge)
                nc->edge.insert({c.t, a.x, a.y, a.z, b.x, b.y, b.z});  // environment.hpp:600-609
            else
                nc->vertex.insert(pack4(c.t, a.x, a.y, a.z));          // environment.hpp:593-599
            child->cons[ag] = nc;
            child->total -= child->cost[ag];
            child->lb -= child->fmin[ag];
            LowLevelResult res;
            bool ok = low_level(g, ag, starts[ag], goals[ag], *nc, child->sol, w, ll_expanded, res);
            if (ok) {
                child->sol[ag] = res.path;
                child->cost[ag] = res.cost;
                child->fmin[ag] = res.fmin;
                child->total += res.cost;
                child->lb += res.fmin;
                child->focal = count_conflicts(g, child->sol);
                all.push_back(child);
                open.insert({child->total, child->id});
                if (child->total <= best_cost * w) focal.insert({child->focal, child->total, child->id});
            } else {
                all.push_back(nullptr);  // keep ids aligned with `all`
            }
        }
        all[id].reset();  // expanded nodes are no longer needed
    }
    if (!goal_node) return 2;

    // ecbs_planner.hpp:34-70
    int makespan = 0, sum = 0;
    for (int i = 0; i < N; ++i) {
        makespan = std::max(makespan, goal_node->cost[i]);
        sum += goal_node->cost[i];
    }
    const int M = makespan + 2;
    out->N = N;
    out->M = M;
    out->makespan = makespan;
    out->sum_cost = sum;
    out->high_level_expanded = hl_expanded;
    out->low_level_expanded = ll_expanded;
    out->T = (double*)malloc(sizeof(double) * (M + 1));
    for (int i = 0; i <= M; ++i) out->T[i] = i * param->time_step;
    out->init_traj = (float*)malloc(sizeof(float) * (size_t)N * (M + 1) * 3);
    for (int a = 0; a < N; ++a) {
        float* tr = out->init_traj + (size_t)a * (M + 1) * 3;
        int n = 0;
        auto push = [&](double x, double y, double z) {
            tr[3 * n] = (float)x, tr[3 * n + 1] = (float)y, tr[3 * n + 2] = (float)z;
            ++n;
        };
        push(mission->start[9 * a], mission->start[9 * a + 1], mission->start[9 * a + 2]);
        for (auto& c : goal_node->sol[a]) push(c.x * gres[0] + gmin[0], c.y * gres[1] + gmin[1], c.z * gres[2] + gmin[2]);
        while (n <= makespan + 2) push(mission->goal[9 * a], mission->goal[9 * a + 1], mission->goal[9 * a + 2]);
    }
    return RBP_OK;
}

extern "C" void rbp_init_traj_free(rbp_init_traj_buf* t) {
    if (!t) return;
    free(t->T);
    free(t->init_traj);
    memset(t, 0, sizeof(*t));
}
