// Host side of the C ABI: instance sources and the fluid solution.
//   - CSV folder reader      <- environments/SO_DFJSP_instance_read.py:6-89
//   - seeded random generator<- environments/Instance_generate.py:19-94
//   - fluid LP driver        <- environments/class_FJSSP.py:246-280 (see fjsp_lp.cpp)
#include "../../include/fjsp_amd.h"
#include "fjsp_host.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>

namespace fjsp {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

int finalize_instance(Instance &in) {
    if (in.R <= 0 || in.M <= 0 || in.S <= 0) { set_error("instance: R, M, S must be positive"); return FJSP_E_FORMAT; }
    if ((int)in.Jr.size() != in.R) { set_error("instance: Jr size"); return FJSP_E_FORMAT; }
    in.koff.assign(in.R + 1, 0);
    for (int r = 0; r < in.R; ++r) {
        if (in.Jr[r] <= 0) { set_error("instance: kind without operations"); return FJSP_E_FORMAT; }
        in.koff[r + 1] = in.koff[r] + in.Jr[r];
    }
    in.K = in.koff[in.R];
    size_t km = (size_t)in.K * in.M;
    if (in.p.size() != km || in.elig_list.size() != km || (int)in.elig_n.size() != in.K) {
        set_error("instance: p / elig sizes"); return FJSP_E_FORMAT;
    }
    if ((int)in.count.size() != in.S * in.R || (int)in.arrive.size() != in.S || (int)in.delivery.size() != in.S) {
        set_error("instance: order sizes"); return FJSP_E_FORMAT;
    }
    for (int k = 0; k < in.K; ++k) {
        int n = 0;
        for (int m = 0; m < in.M; ++m) {
            int v = in.p[(size_t)k * in.M + m];
            if (v < 0) { set_error("instance: negative processing time"); return FJSP_E_FORMAT; }
            if (v > 0) ++n;
        }
        if (n == 0 || n != in.elig_n[k]) { set_error("instance: eligibility list inconsistent with p"); return FJSP_E_FORMAT; }
        for (int i = 0; i < n; ++i) {
            int m = in.elig_list[(size_t)k * in.M + i];
            if (m < 0 || m >= in.M || in.p[(size_t)k * in.M + m] <= 0) { set_error("instance: bad eligible machine"); return FJSP_E_FORMAT; }
        }
    }
    for (int v : in.count)
        if (v <= 0) { set_error("instance: count_sr must be >= 1 (reference divides by it, class_FJSSP.py:214)"); return FJSP_E_FORMAT; }
    in.x.assign(km, 0.0);
    in.has_x = false;
    in.valid = true;
    return FJSP_OK;
}

// ---- csv.reader subset: comma separated, double-quoted fields -------------
static bool read_csv(const std::string &file, std::vector<std::vector<std::string>> &rows) {
    std::ifstream f(file);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::vector<std::string> row;
        std::string cur;
        bool q = false;
        for (size_t i = 0; i < line.size(); ++i) {
            char c = line[i];
            if (q) {
                if (c == '"') {
                    if (i + 1 < line.size() && line[i + 1] == '"') { cur.push_back('"'); ++i; }
                    else q = false;
                } else cur.push_back(c);
            } else if (c == '"') q = true;
            else if (c == ',') { row.push_back(cur); cur.clear(); }
            else cur.push_back(c);
        }
        row.push_back(cur);
        if (row.size() == 1 && row[0].empty()) continue;  // csv.reader skips blank lines
        rows.push_back(row);
    }
    return true;
}
// re.findall(r'\d+', s) -> ints  (SO_DFJSP_instance_read.py:31-39)
static std::vector<int> digits(const std::string &s) {
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size()) {
        if (s[i] >= '0' && s[i] <= '9') {
            long v = 0;
            while (i < s.size() && s[i] >= '0' && s[i] <= '9') { v = v * 10 + (s[i] - '0'); ++i; }
            out.push_back((int)v);
        } else ++i;
    }
    return out;
}
static bool first_int(const std::string &s, int &v) {
    auto d = digits(s);
    if (d.empty()) return false;
    v = d[0];
    return true;
}

static int load_csv(Instance &in, const std::string &dir) {
    std::vector<std::vector<std::string>> rows;
    // based_data.csv  (:49-54)
    if (!read_csv(dir + "/based_data.csv", rows) || rows.size() < 2 || rows[1].size() < 4) {
        set_error("cannot read " + dir + "/based_data.csv (need kind_count,machine_count,order_count,DDT)");
        return FJSP_E_IO;
    }
    int ddt_i = 0;
    if (!first_int(rows[1][0], in.R) || !first_int(rows[1][1], in.M) || !first_int(rows[1][2], in.S) ||
        !first_int(rows[1][3], ddt_i)) { set_error("based_data.csv: non-numeric field"); return FJSP_E_FORMAT; }
    in.ddt = (double)ddt_i;  // `\d+` extraction: "0.5" -> 0, "1.5" -> 1 (:36-40,53)
    // order_data.csv (:78-89)
    rows.clear();
    if (!read_csv(dir + "/order_data.csv", rows) || (int)rows.size() < 1 + in.S) {
        set_error("cannot read " + dir + "/order_data.csv"); return FJSP_E_IO;
    }
    in.count.assign((size_t)in.S * in.R, 0);
    in.arrive.assign(in.S, 0);
    in.delivery.assign(in.S, 0);
    std::vector<char> seen(in.S, 0);
    for (size_t i = 1; i < rows.size(); ++i) {
        if (rows[i].size() < 4) { set_error("order_data.csv: short row"); return FJSP_E_FORMAT; }
        int s, a, d;
        if (!first_int(rows[i][0], s) || !first_int(rows[i][1], a) || !first_int(rows[i][2], d)) {
            set_error("order_data.csv: non-numeric field"); return FJSP_E_FORMAT;
        }
        auto c = digits(rows[i][3]);
        if (s < 0 || s >= in.S || (int)c.size() < in.R) { set_error("order_data.csv: bad order row"); return FJSP_E_FORMAT; }
        for (int r = 0; r < in.R; ++r) in.count[(size_t)s * in.R + r] = c[r];
        in.arrive[s] = a; in.delivery[s] = d; seen[s] = 1;
    }
    for (int s = 0; s < in.S; ++s) if (!seen[s]) { set_error("order_data.csv: missing order"); return FJSP_E_FORMAT; }
    // process_data.csv (:55-76)
    rows.clear();
    if (!read_csv(dir + "/process_data.csv", rows) || rows.size() < 2) {
        set_error("cannot read " + dir + "/process_data.csv"); return FJSP_E_IO;
    }
    in.Jr.assign(in.R, 0);
    struct Row { int r, j; std::vector<int> ms, ts, ps; };
    std::vector<Row> prow;
    for (size_t i = 1; i < rows.size(); ++i) {
        if (rows[i].size() < 4) { set_error("process_data.csv: short row"); return FJSP_E_FORMAT; }
        Row w;
        if (!first_int(rows[i][0], w.r) || !first_int(rows[i][1], w.j)) { set_error("process_data.csv: non-numeric"); return FJSP_E_FORMAT; }
        w.ms = digits(rows[i][2]); w.ts = digits(rows[i][3]);
        if (rows[i].size() >= 5) w.ps = digits(rows[i][4]);     // power column (MO_DFJSP_instance_read.py:84)
        if (w.r < 0 || w.r >= in.R) { set_error("process_data.csv: kind out of range"); return FJSP_E_FORMAT; }
        // task labels must be 0..J_r-1 in file order: the LP's (r, j+1) arithmetic
        // (class_FJSSP.py:270-271) assumes it.
        if (w.j != in.Jr[w.r]) { set_error("process_data.csv: operations of a kind must be numbered 0..J-1 in order"); return FJSP_E_FORMAT; }
        in.Jr[w.r]++;
        prow.push_back(w);
    }
    in.koff.assign(in.R + 1, 0);
    for (int r = 0; r < in.R; ++r) in.koff[r + 1] = in.koff[r] + in.Jr[r];
    in.K = in.koff[in.R];
    size_t km = (size_t)in.K * in.M;
    in.p.assign(km, 0); in.elig_list.assign(km, 0); in.elig_n.assign(in.K, 0);
    for (const Row &w : prow) {
        int k = in.koff[w.r] + w.j;
        // zip(machines, times) into a dict (:72-75): later duplicates overwrite, order of first insertion kept
        size_t n = std::min(w.ms.size(), w.ts.size());
        for (size_t q = 0; q < n; ++q) {
            int m = w.ms[q];
            if (m < 0 || m >= in.M) { set_error("process_data.csv: machine out of range"); return FJSP_E_FORMAT; }
            if (w.ts[q] <= 0) { set_error("process_data.csv: processing time must be positive"); return FJSP_E_FORMAT; }
            in.p[(size_t)k * in.M + m] = w.ts[q];
        }
        // machine_rj_dict[(r,j)] keeps the FILE tuple (duplicates would be kept by the
        // reference; reject them, the set-order emulation assumes distinct machines)
        for (size_t q = 0; q < w.ms.size(); ++q) {
            for (size_t q2 = 0; q2 < q; ++q2)
                if (w.ms[q2] == w.ms[q]) { set_error("process_data.csv: duplicate machine in a row"); return FJSP_E_FORMAT; }
            if (q >= w.ts.size()) { set_error("process_data.csv: machine without time"); return FJSP_E_FORMAT; }
            in.elig_list[(size_t)k * in.M + in.elig_n[k]++] = w.ms[q];
        }
    }
    // machine_data.csv (MO_DFJSP_instance_read.py:56-73): idle power per machine and breakdown windows;
    // present only in the dynamic multi-objective folders (data/HMPSAC, data/industrial)
    std::vector<std::vector<std::string>> mrows;
    if (read_csv(dir + "/machine_data.csv", mrows) && mrows.size() >= 2) {
        in.has_dynamic = true;
        in.power.assign(km, 0);
        for (const Row &w : prow) {
            int k = in.koff[w.r] + w.j;
            for (size_t q = 0; q < std::min(w.ms.size(), w.ps.size()); ++q) in.power[(size_t)k * in.M + w.ms[q]] = w.ps[q];
        }
        in.idle_power.assign(in.M, -1);
        std::vector<std::vector<int>> win(in.M);
        for (size_t i = 1; i < mrows.size(); ++i) {
            int m, ip;
            if (mrows[i].size() < 2 || !first_int(mrows[i][0], m) || !first_int(mrows[i][1], ip) || m < 0 || m >= in.M) {
                set_error("machine_data.csv: bad row"); return FJSP_E_FORMAT;
            }
            if (in.idle_power[m] < 0) in.idle_power[m] = ip;          // first row of a machine sets the idle power (:65-66)
            if (mrows[i].size() > 2) {                                // breakdown window (:68-71)
                int bs, be;
                if (mrows[i].size() < 4 || !first_int(mrows[i][2], bs) || !first_int(mrows[i][3], be)) {
                    set_error("machine_data.csv: bad breakdown window"); return FJSP_E_FORMAT;
                }
                win[m].push_back(bs); win[m].push_back(be);
            }
        }
        in.bk_n.assign(in.M, 0);
        for (int m = 0; m < in.M; ++m) {
            if (in.idle_power[m] < 0) { set_error("machine_data.csv: machine without a row"); return FJSP_E_FORMAT; }
            in.bk_n[m] = (int)win[m].size() / 2;
            in.bk.insert(in.bk.end(), win[m].begin(), win[m].end());
        }
    }
    {
        const bool dyn = in.has_dynamic;
        std::vector<int> power = in.power, idle = in.idle_power, bkn = in.bk_n, bk = in.bk;
        int rc = finalize_instance(in);
        in.has_dynamic = dyn; in.power = power; in.idle_power = idle; in.bk_n = bkn; in.bk = bk;
        return rc;
    }
}

// ---- counter-based generator ----------------------------------------------
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    int randint(int a, int b) { return a + (int)(((next() >> 32) * (uint64_t)(b - a + 1)) >> 32); }
    double uniform(double a, double b) { return a + (b - a) * ((double)(next() >> 11) * (1.0 / 9007199254740992.0)); }
};

static int generate(Instance &in, uint64_t seed, const fjsp_gen_params &g) {
    if (g.M <= 0 || g.S <= 0 || g.R_min <= 0 || g.R_max < g.R_min || g.J_min <= 0 || g.J_max < g.J_min ||
        g.p_min <= 0 || g.p_max < g.p_min || g.N_min <= 0 || g.N_max < g.N_min) {
        set_error("generator: bad parameters"); return FJSP_E_ARG;
    }
    Rng rng(seed);
    in = Instance();
    in.ddt = g.DDT; in.M = g.M; in.S = g.S;
    in.R = rng.randint(g.R_min, g.R_max);                                    // :42
    in.Jr.resize(in.R);
    for (int r = 0; r < in.R; ++r) in.Jr[r] = rng.randint(g.J_min, g.J_max); // :46,71
    in.koff.assign(in.R + 1, 0);
    for (int r = 0; r < in.R; ++r) in.koff[r + 1] = in.koff[r] + in.Jr[r];
    in.K = in.koff[in.R];
    size_t km = (size_t)in.K * in.M;
    in.p.assign(km, 0); in.elig_list.assign(km, 0); in.elig_n.assign(in.K, 0);
    std::vector<int> perm(in.M);
    for (int k = 0; k < in.K; ++k) {                                         // :73 choice(M, U{1..M}, replace=False)
        int n = rng.randint(1, in.M);
        for (int m = 0; m < in.M; ++m) perm[m] = m;
        for (int i = 0; i < n; ++i) {
            int j = rng.randint(i, in.M - 1);
            std::swap(perm[i], perm[j]);
            in.elig_list[(size_t)k * in.M + i] = perm[i];
        }
        in.elig_n[k] = n;
    }
    for (int k = 0; k < in.K; ++k)                                           // :74
        for (int i = 0; i < in.elig_n[k]; ++i)
            in.p[(size_t)k * in.M + in.elig_list[(size_t)k * in.M + i]] = rng.randint(g.p_min, g.p_max);
    std::vector<double> time_rj(in.K);                                       // :78 mean over machine_rj_dict order
    for (int k = 0; k < in.K; ++k) {
        long s = 0;
        for (int i = 0; i < in.elig_n[k]; ++i) s += in.p[(size_t)k * in.M + in.elig_list[(size_t)k * in.M + i]];
        time_rj[k] = (double)s / (double)in.elig_n[k];
    }
    in.count.assign((size_t)in.S * in.R, 0);
    for (int s = 0; s < in.S; ++s)
        for (int r = 0; r < in.R; ++r) in.count[(size_t)s * in.R + r] = rng.randint(g.N_min, g.N_max); // :80
    std::vector<double> gap(in.S), interval(in.S, 0.0);
    for (int s = 0; s < in.S; ++s) {                                         // :81-82
        double acc = 0.0;
        for (int r = 0; r < in.R; ++r)
            for (int j = 0; j < in.Jr[r]; ++j) acc = acc + time_rj[in.koff[r] + j] * (double)in.count[(size_t)s * in.R + r];
        gap[s] = acc * g.DDT / (double)(in.M * 2);
    }
    for (int s = 1; s < in.S; ++s) interval[s] = rng.uniform(g.t_si_min, g.t_si_max); // :83-84
    in.arrive.assign(in.S, 0); in.delivery.assign(in.S, 0);
    std::vector<double> dl(in.S);
    for (int s = 0; s < in.S; ++s) {                                         // :85-86
        double acc = 0.0;
        for (int q = 0; q <= s; ++q) acc = acc + interval[q];
        in.arrive[s] = (int)acc;
        dl[s] = (double)in.arrive[s] + gap[s];
    }
    std::sort(dl.begin(), dl.end());                                         // :87
    for (int s = 0; s < in.S; ++s) in.delivery[s] = (int)dl[s];              // :88
    return finalize_instance(in);
}

static int solve_order0(Instance &in) {
    std::vector<int> Q(in.K), now(in.K);
    for (int r = 0; r < in.R; ++r)
        for (int j = 0; j < in.Jr[r]; ++j) {
            Q[in.koff[r] + j] = in.count[r];
            now[in.koff[r] + j] = (j == 0) ? in.count[r] : 0;
        }
    int rc = solve_fluid_lp(in.R, in.M, in.Jr.data(), in.p.data(), Q.data(), now.data(), in.x.data(), &in.lp_objective);
    if (rc != 0) return FJSP_E_LP;
    in.has_x = true;
    return FJSP_OK;
}

}  // namespace fjsp

using namespace fjsp;

extern "C" {

const char *fjsp_last_error(void) { return g_err.c_str(); }
int fjsp_abi_version(void) { return FJSP_ABI_VERSION; }

int fjsp_instances_create(int32_t n, fjsp_instances **out) {
    if (n <= 0 || !out) { set_error("fjsp_instances_create: bad arguments"); return FJSP_E_ARG; }
    auto *s = new fjsp_instances();
    s->v.resize((size_t)n);
    *out = s;
    return FJSP_OK;
}
void fjsp_instances_destroy(fjsp_instances *s) { delete s; }
int fjsp_instances_count(const fjsp_instances *s) { return s ? (int)s->v.size() : 0; }

#define CHECK_IDX(s, i) \
    if (!(s) || (i) < 0 || (size_t)(i) >= (s)->v.size()) { set_error("instance index out of range"); return FJSP_E_ARG; }

int fjsp_instances_load_csv(fjsp_instances *s, int32_t i, const char *path, const char *file_name) {
    CHECK_IDX(s, i);
    if (!path || !file_name) { set_error("load_csv: null path"); return FJSP_E_ARG; }
    Instance in;
    int rc = load_csv(in, std::string(path) + "/" + file_name);
    if (rc == FJSP_OK) s->v[(size_t)i] = std::move(in);
    return rc;
}

int fjsp_instances_generate(fjsp_instances *s, int32_t i, uint64_t seed, const fjsp_gen_params *prm) {
    CHECK_IDX(s, i);
    if (!prm) { set_error("generate: null params"); return FJSP_E_ARG; }
    Instance in;
    int rc = generate(in, seed, *prm);
    if (rc == FJSP_OK) s->v[(size_t)i] = std::move(in);
    return rc;
}

int fjsp_instances_set_raw(fjsp_instances *s, int32_t i, int32_t R, int32_t M, int32_t S, const int32_t *Jr,
                           const int32_t *p, const int32_t *elig_n, const int32_t *elig_list, const int32_t *count,
                           const int32_t *arrive, const int32_t *delivery, double ddt) {
    CHECK_IDX(s, i);
    if (R <= 0 || M <= 0 || S <= 0 || !Jr || !p || !elig_n || !elig_list || !count || !arrive || !delivery) {
        set_error("set_raw: bad arguments"); return FJSP_E_ARG;
    }
    Instance in;
    in.R = R; in.M = M; in.S = S; in.ddt = ddt;
    in.Jr.assign(Jr, Jr + R);
    int K = 0;
    for (int r = 0; r < R; ++r) { if (Jr[r] <= 0) { set_error("set_raw: Jr"); return FJSP_E_FORMAT; } K += Jr[r]; }
    in.p.assign(p, p + (size_t)K * M);
    in.elig_n.assign(elig_n, elig_n + K);
    in.elig_list.assign(elig_list, elig_list + (size_t)K * M);
    in.count.assign(count, count + (size_t)S * R);
    in.arrive.assign(arrive, arrive + S);
    in.delivery.assign(delivery, delivery + S);
    int rc = finalize_instance(in);
    if (rc == FJSP_OK) s->v[(size_t)i] = std::move(in);
    return rc;
}

int fjsp_instances_dims(const fjsp_instances *s, int32_t i, int32_t dims[6]) {
    CHECK_IDX(s, i);
    const Instance &in = s->v[(size_t)i];
    if (!in.valid) { set_error("instance not populated"); return FJSP_E_STATE; }
    dims[0] = in.R; dims[1] = in.M; dims[2] = in.K; dims[3] = in.S;
    dims[4] = in.jobs_of_order(0); dims[5] = in.jobs_total();
    return FJSP_OK;
}

int fjsp_instances_get(const fjsp_instances *s, int32_t i, int32_t *Jr, int32_t *p, int32_t *elig_n,
                       int32_t *elig_list, int32_t *count, int32_t *arrive, int32_t *delivery, double *ddt, double *x) {
    CHECK_IDX(s, i);
    const Instance &in = s->v[(size_t)i];
    if (!in.valid) { set_error("instance not populated"); return FJSP_E_STATE; }
    if (Jr) std::copy(in.Jr.begin(), in.Jr.end(), Jr);
    if (p) std::copy(in.p.begin(), in.p.end(), p);
    if (elig_n) std::copy(in.elig_n.begin(), in.elig_n.end(), elig_n);
    if (elig_list) std::copy(in.elig_list.begin(), in.elig_list.end(), elig_list);
    if (count) std::copy(in.count.begin(), in.count.end(), count);
    if (arrive) std::copy(in.arrive.begin(), in.arrive.end(), arrive);
    if (delivery) std::copy(in.delivery.begin(), in.delivery.end(), delivery);
    if (ddt) *ddt = in.ddt;
    if (x) std::copy(in.x.begin(), in.x.end(), x);
    return FJSP_OK;
}

int fjsp_instances_dynamic_dims(const fjsp_instances *s, int32_t i, int32_t dims[2]) {
    CHECK_IDX(s, i);
    const Instance &in = s->v[(size_t)i];
    if (!in.valid) { set_error("instance not populated"); return FJSP_E_STATE; }
    dims[0] = in.has_dynamic ? 1 : 0;
    dims[1] = (int)in.bk.size() / 2;
    return FJSP_OK;
}

int fjsp_instances_get_dynamic(const fjsp_instances *s, int32_t i, int32_t *power, int32_t *idle_power, int32_t *bk_n,
                               int32_t *bk) {
    CHECK_IDX(s, i);
    const Instance &in = s->v[(size_t)i];
    if (!in.valid || !in.has_dynamic) { set_error("instance has no machine data"); return FJSP_E_STATE; }
    if (power) std::copy(in.power.begin(), in.power.end(), power);
    if (idle_power) std::copy(in.idle_power.begin(), in.idle_power.end(), idle_power);
    if (bk_n) std::copy(in.bk_n.begin(), in.bk_n.end(), bk_n);
    if (bk) std::copy(in.bk.begin(), in.bk.end(), bk);
    return FJSP_OK;
}

int fjsp_instances_set_dynamic(fjsp_instances *s, int32_t i, const int32_t *power, const int32_t *idle_power,
                               const int32_t *bk_n, const int32_t *bk) {
    CHECK_IDX(s, i);
    Instance &in = s->v[(size_t)i];
    if (!in.valid || !power || !idle_power || !bk_n) { set_error("set_dynamic: bad arguments"); return FJSP_E_ARG; }
    in.power.assign(power, power + (size_t)in.K * in.M);
    in.idle_power.assign(idle_power, idle_power + in.M);
    in.bk_n.assign(bk_n, bk_n + in.M);
    int total = 0;
    for (int m = 0; m < in.M; ++m) { if (bk_n[m] < 0) { set_error("set_dynamic: negative window count"); return FJSP_E_ARG; } total += bk_n[m]; }
    if (total > 0 && !bk) { set_error("set_dynamic: windows missing"); return FJSP_E_ARG; }
    in.bk.assign(bk, bk + (size_t)total * 2);
    in.has_dynamic = true;
    return FJSP_OK;
}

int fjsp_instances_solve_fluid(fjsp_instances *s, int32_t first, int32_t n, int32_t n_threads) {
    if (!s || first < 0 || n <= 0 || (size_t)first + (size_t)n > s->v.size()) { set_error("solve_fluid: range"); return FJSP_E_ARG; }
    for (int i = first; i < first + n; ++i)
        if (!s->v[(size_t)i].valid) { set_error("solve_fluid: instance not populated"); return FJSP_E_STATE; }
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    if (n_threads > n) n_threads = n;
    std::atomic<int> next(first), fail(0);
    std::string err;
    auto work = [&]() {
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= first + n) break;
            if (solve_order0(s->v[(size_t)i]) != FJSP_OK) {
                if (fail.fetch_add(1) == 0) err = g_err;  // thread-local message of the worker
            }
        }
    };
    if (n_threads == 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
    if (fail.load()) { set_error(err.empty() ? "fluid LP failed" : err); return FJSP_E_LP; }
    return FJSP_OK;
}

int fjsp_instances_set_x(fjsp_instances *s, int32_t i, const double *x) {
    CHECK_IDX(s, i);
    Instance &in = s->v[(size_t)i];
    if (!in.valid || !x) { set_error("set_x: bad arguments"); return FJSP_E_ARG; }
    for (int k = 0; k < in.K; ++k) {
        double acc = 0.0;
        for (int m = 0; m < in.M; ++m) {
            double v = x[(size_t)k * in.M + m];
            if (in.p[(size_t)k * in.M + m] == 0) { if (v != 0.0) { set_error("set_x: nonzero x on an ineligible pair"); return FJSP_E_ARG; } }
            else acc += v / (double)in.p[(size_t)k * in.M + m];
        }
        if (!(acc > 0.0)) { set_error("set_x: operation type with zero fluid rate"); return FJSP_E_ARG; }
    }
    std::copy(x, x + (size_t)in.K * in.M, in.x.begin());
    in.has_x = true;
    return FJSP_OK;
}

int fjsp_fluid_lp(int32_t R, int32_t M, const int32_t *Jr, const int32_t *p, const int32_t *Q, const int32_t *n_now,
                  double *x, double *objective) {
    if (R <= 0 || M <= 0 || !Jr || !p || !Q || !n_now || !x) { set_error("fjsp_fluid_lp: bad arguments"); return FJSP_E_ARG; }
    return solve_fluid_lp(R, M, Jr, p, Q, n_now, x, objective) == 0 ? FJSP_OK : FJSP_E_LP;
}

}  // extern "C"
