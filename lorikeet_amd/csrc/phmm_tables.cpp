// Host-side construction of the quality -> probability tables that live in HBM.
//
// The reference rebuilds these per PairHMM, i.e. once per assembly region
// (PairHMMModel::new, src/pair_hmm/pair_hmm_model.rs:47-78) and evaluates
// 10^(-q/10) with powf per matrix cell (pair_hmm.rs:638-654).  Here they are built once per
// process with the same libm calls, in the same order, so every entry is bit-identical to what
// the reference's scalar path computes, then uploaded once per handle.
#include "phmm_tables.hpp"

#include <algorithm>
#include <cmath>
#include <mutex>

namespace phmm {
namespace {

constexpr double kMaxTolerance = 8.0;  // JacobianLogTable::MAX_TOLERANCE (math_utils.rs:485)
constexpr double kTableStep = 0.0001;  // JacobianLogTable::TABLE_STEP   (math_utils.rs:490)

struct Tables {
    std::vector<double> jacobian;  // log10(1 + 10^(-k*step))            (math_utils.rs:9-14)
    std::vector<double> eps;       // 10^(-q/10), q = 0..255              (quality_utils.rs:98-104)
    std::vector<double> eps_third; // eps / 3.0  (TRISTATE_CORRECTION)    (pair_hmm.rs:53,646-651)
    std::vector<double> mm;        // match->match, triangular, 0..=255   (pair_hmm_model.rs:47-78,442-461)
};

double approx_log10_sum_log10(const Tables &t, double a, double b) {  // math_utils.rs:314-332
    if (a > b) std::swap(a, b);
    if (a == -INFINITY) return b;
    const double diff = b - a;
    if (!(diff < kMaxTolerance)) return b;
    const double inv_step = 1.0 / kTableStep;
    return b + t.jacobian[(size_t)std::round(diff * inv_step)];
}

const Tables &tables() {
    static Tables t;
    static std::once_flag once;
    std::call_once(once, [] {
        const size_t nj = (size_t)((kMaxTolerance / kTableStep) + 1.0);
        t.jacobian.resize(nj);
        for (size_t k = 0; k < nj; ++k) t.jacobian[k] = std::log10(1.0 + std::pow(10.0, -((double)k) * kTableStep));

        t.eps.resize(256);
        t.eps_third.resize(256);
        for (int q = 0; q < 256; ++q) {
            t.eps[q] = std::pow(10.0, ((double)q) / -10.0);
            t.eps_third[q] = t.eps[q] / 3.0;
        }

        // Rows 0..=254 are the reference's lookup table; row 255 holds what its direct formula
        // (pair_hmm_model.rs:453-457, max_qual > MAX_QUAL) returns, so the device needs no branch.
        const double inv_ln10 = 1.0 / std::log(10.0);
        t.mm.resize((size_t)256 * 257 / 2);
        size_t offset = 0;
        for (int i = 0; i < 256; ++i) {
            for (int j = 0; j <= i; ++j) {
                const double log10_sum = approx_log10_sum_log10(t, -0.1 * (double)i, -0.1 * (double)j);
                const double p = std::pow(10.0, log10_sum);
                if (i <= 254) {
                    const double l10 = std::log1p(-std::fmin(1.0, p)) * inv_ln10;
                    t.mm[offset + j] = std::pow(10.0, l10);
                } else {
                    t.mm[offset + j] = 1.0 - p;
                }
            }
            offset += (size_t)i + 1;
        }
    });
    return t;
}

}  // namespace

const std::vector<double> &table_eps() { return tables().eps; }
const std::vector<double> &table_eps_third() { return tables().eps_third; }
const std::vector<double> &table_match_to_match() { return tables().mm; }
double initial_condition() { return std::pow(2.0, 1020.0); }                       // pair_hmm.rs:16
double initial_condition_log10() { return std::log10(std::pow(2.0, 1020.0)); }    // pair_hmm.rs:17


// PairHMMLikelihoodCalculationEngine::initialize_pcr_error_model / get_error_model_adjusted_qual
// (pair_hmm_likelihood_calculation_engine.rs:169-193): max(6, (40 - exp(len / (rate * pi)) + 1) as usize) as u8,
// where the rate factor is the enum discriminant of the PCR model.
std::vector<unsigned char> pcr_error_model_cache(int model) {
    std::vector<unsigned char> cache(101, 0);
    if (model <= 0) return cache;
    const double pi = 3.14159265358979323846264338327950288;  // std::f64::consts::PI
    for (int i = 0; i <= 100; ++i) {
        const double v = 40.0 - std::exp((double)i / ((double)model * pi)) + 1.0;
        const long u = v > 0.0 ? (long)v : 0;  // Rust float -> usize casts saturate at 0
        cache[i] = (unsigned char)std::max<long>(6, u);
    }
    return cache;
}

}  // namespace phmm
