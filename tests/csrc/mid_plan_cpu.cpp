// Test infrastructure: the planner of the mid-size factorisation schedule (madqp_jl_amd/csrc/mid_plan.inc) behind a C
// entry point, so that tests/test_mid_plan.py can check its invariants without a GPU.
#include "../../madqp_jl_amd/csrc/mid_plan.inc"

// Plans nblk block steps on `cap` tiles per round; returns the number of rounds per step the planner settled on (0:
// no plan) and writes up to max_rows rows (step, tile row, tile column, first panel, panels), one per unit in the order
// of the step's list, to out; *nrows = rows in all; units_out[k] = units of step k.
extern "C" int mid_plan_rows(int nblk, int cap, int32_t* out, int64_t max_rows, int64_t* nrows, int32_t* units_out) {
    std::vector<std::vector<MidVisit>> steps;
    std::vector<int32_t> units;
    int rounds = 0;
    for (int r = 1; r <= 64; ++r)
        if (mid_plan_steps(nblk, r * cap, steps, units, cap)) {
            rounds = r;
            break;
        }
    *nrows = 0;
    if (!rounds) return 0;
    for (int k = 0; k < nblk; ++k) {
        std::vector<uint32_t> w;
        mid_plan_units(nblk, k, steps[k], w);
        if (units_out) units_out[k] = units[k];
        if ((int)w.size() != units[k]) return -1;
        for (uint32_t e : w) {
            if (*nrows < max_rows) {
                int32_t* row = out + 5 * *nrows;
                row[0] = k, row[1] = e & 255, row[2] = (e >> 8) & 255, row[3] = (e >> 16) & 255, row[4] = e >> 24;
            }
            ++*nrows;
        }
    }
    return rounds;
}
