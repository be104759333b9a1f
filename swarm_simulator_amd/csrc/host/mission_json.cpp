// Mission JSON reader — host front-end, mirrors Mission::setMission
// (reference: swarm_planner/include/mission.hpp:22-88).  The reference parses with rapidjson; this is a
// self-contained recursive-descent parser for the subset of JSON the mission files use.
#include "rbp_host.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;  // keeps file order
    const JValue* find(const std::string& k) const {
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct Parser {
    const char* p;
    const char* end;
    bool ok = true;
    void ws() {
        while (p < end && (isspace((unsigned char)*p))) ++p;
    }
    bool lit(const char* s) {
        size_t n = strlen(s);
        if ((size_t)(end - p) >= n && !strncmp(p, s, n)) {
            p += n;
            return true;
        }
        return false;
    }
    JValue value() {
        JValue v;
        ws();
        if (p >= end) {
            ok = false;
            return v;
        }
        if (*p == '{') {
            ++p;
            v.kind = JValue::Obj;
            ws();
            if (p < end && *p == '}') {
                ++p;
                return v;
            }
            while (ok) {
                ws();
                JValue k = value();
                if (k.kind != JValue::Str) {
                    ok = false;
                    break;
                }
                ws();
                if (p >= end || *p != ':') {
                    ok = false;
                    break;
                }
                ++p;
                v.obj.emplace_back(k.str, value());
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    break;
                }
                ok = false;
            }
        } else if (*p == '[') {
            ++p;
            v.kind = JValue::Arr;
            ws();
            if (p < end && *p == ']') {
                ++p;
                return v;
            }
            while (ok) {
                v.arr.push_back(value());
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    break;
                }
                ok = false;
            }
        } else if (*p == '"') {
            ++p;
            v.kind = JValue::Str;
            while (p < end && *p != '"') {
                if (*p == '\\' && p + 1 < end) {
                    ++p;
                    switch (*p) {
                        case 'n': v.str += '\n'; break;
                        case 't': v.str += '\t'; break;
                        default: v.str += *p;
                    }
                    ++p;
                } else
                    v.str += *p++;
            }
            if (p >= end)
                ok = false;
            else
                ++p;
        } else if (lit("true")) {
            v.kind = JValue::Bool;
            v.b = true;
        } else if (lit("false")) {
            v.kind = JValue::Bool;
        } else if (lit("null")) {
        } else {
            char* e = nullptr;
            v.num = strtod(p, &e);
            if (e == p || e > end)
                ok = false;
            else {
                v.kind = JValue::Num;
                p = e;
            }
        }
        return v;
    }
};

void fill(const JValue* a, double* dst, int cap) {
    // mission.hpp:47-53: state(9,0) then the first start.Size() entries overwritten
    if (!a || a->kind != JValue::Arr) return;
    for (size_t i = 0; i < a->arr.size() && (int)i < cap; ++i) dst[i] = a->arr[i].num;
}

}  // namespace

extern "C" int rbp_mission_load_json(const char* path, rbp_mission_buf* out) {
    if (!path || !out) return RBP_ERR_BAD_ARGUMENT;
    memset(out, 0, sizeof(*out));
    FILE* f = fopen(path, "rb");
    if (!f) return RBP_ERR_BAD_ARGUMENT;
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    fclose(f);
    Parser ps{text.data(), text.data() + text.size()};
    JValue doc = ps.value();
    if (!ps.ok || doc.kind != JValue::Obj) return RBP_ERR_BAD_ARGUMENT;
    const JValue* agents = doc.find("agents");
    const JValue* quads = doc.find("quadrotors");
    if (!agents || agents->kind != JValue::Arr || !quads) return RBP_ERR_BAD_ARGUMENT;
    int N = (int)agents->arr.size();
    out->N = N;
    out->start = (double*)calloc((size_t)N * 9, sizeof(double));
    out->goal = (double*)calloc((size_t)N * 9, sizeof(double));
    out->radius = (double*)calloc(N, sizeof(double));
    out->speed = (double*)calloc(N, sizeof(double));
    out->max_vel = (double*)calloc((size_t)N * 3, sizeof(double));
    out->max_acc = (double*)calloc((size_t)N * 3, sizeof(double));
    for (int qi = 0; qi < N; ++qi) {
        const JValue& ag = agents->arr[qi];
        const JValue* name = ag.find("name");
        fill(ag.find("start"), out->start + 9 * qi, 9);
        fill(ag.find("goal"), out->goal + 9 * qi, 9);
        if (const JValue* r = ag.find("radius")) out->radius[qi] = r->num;  // mission.hpp:64
        if (const JValue* s = ag.find("speed")) out->speed[qi] = s->num;    // mission.hpp:67
        const JValue* q = name ? quads->find(name->str) : nullptr;          // mission.hpp:71
        if (!q) {
            rbp_mission_free(out);
            return RBP_ERR_BAD_ARGUMENT;
        }
        fill(q->find("max_vel"), out->max_vel + 3 * qi, 3);
        fill(q->find("max_acc"), out->max_acc + 3 * qi, 3);
    }
    return RBP_OK;
}

extern "C" void rbp_mission_free(rbp_mission_buf* m) {
    if (!m) return;
    free(m->start);
    free(m->goal);
    free(m->radius);
    free(m->speed);
    free(m->max_vel);
    free(m->max_acc);
    memset(m, 0, sizeof(*m));
}
