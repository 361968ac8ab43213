// Internal host-side types shared by the instance loader, the fluid LP and the
// device batch packer.  Not part of the public ABI (see include/fjsp_amd.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace fjsp {

// One problem instance in the array form of SURVEY.md Appendix A.
// k = koff[r] + j, r-major (kind_task_tuple order).
struct Instance {
    int R = 0, M = 0, K = 0, S = 0;
    std::vector<int> Jr, koff;        // [R], [R+1]
    std::vector<int> p;               // [K*M] k-major, 0 = ineligible
    std::vector<int> elig_n;          // [K]
    std::vector<int> elig_list;       // [K*M] machine_rj_dict[(r,j)] in file order
    std::vector<int> count;           // [S*R]
    std::vector<int> arrive, delivery;// [S]
    // dynamic multi-objective instances (MO_DFJSP_instance_read.py): powers and machine breakdown windows
    bool has_dynamic = false;
    std::vector<int> power;           // [K*M] power_mrj_dict (0 where ineligible)
    std::vector<int> idle_power;      // [M]   power_m_dict
    std::vector<int> bk_n;            // [M]   number of breakdown windows of machine m
    std::vector<int> bk;              // flattened (start, end) pairs, machine-major, file order; bk_off[m] = prefix of bk_n
    double ddt = 0.0;                 // self.DDT as the source parsed it
    std::vector<double> x;            // [K*M] fluid solution (input of the kernels)
    bool has_x = false;
    double lp_objective = 0.0;
    bool valid = false;

    int jobs_of_order(int s) const {
        int n = 0;
        for (int r = 0; r < R; ++r) n += count[(size_t)s * R + r];
        return n;
    }
    int jobs_total() const {
        int n = 0;
        for (int s = 0; s < S; ++s) n += jobs_of_order(s);
        return n;
    }
};

void set_error(const std::string &msg);

// fjsp_lp.cpp -- fluid_model (class_FJSSP.py:246-280) as a dense lexicographic simplex.
int solve_fluid_lp(int R, int M, const int *Jr, const int *p, const int *Q, const int *n_now,
                   double *x, double *objective);

// fjsp_instance.cpp
int finalize_instance(Instance &in);  // derives koff/K, validates

}  // namespace fjsp

// on-policy rollout buffer (fjsp_rollout_buffer.hip); fjsp_env_rollout_policy (fjsp_env.hip) writes its rows in place
struct fjsp_rollout {
    int T = 0, N = 0, S = 0, device = 0, len = 0;
    float *states = nullptr, *actions = nullptr, *rewards = nullptr, *next_states = nullptr, *dones = nullptr,
          *valid = nullptr, *returns = nullptr;
};

struct fjsp_instances {
    std::vector<fjsp::Instance> v;
};
