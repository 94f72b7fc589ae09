// The reference's own PairHMM tests, restated against the C++ host layer
// (lorikeet_amd/csrc/host/lorikeet_pair_hmm.hpp) above the C ABI -- every likelihood below is computed by
// the gfx950 kernels.  Each test names the Rust test it mirrors; tolerances are the reference's.
//
//   reference tests/vector_pair_hmm_unit_tests.rs                         test_likelihoods_avx
//   reference tests/pair_hmm_unit_tests.rs                                make_basic_likelihood_tests, ...
//   reference tests/pair_hmm_likelihood_calculation_engine_unit_tests.rs  test_compute_likelihoods
//
// usage: reference_tests <path to pairhmm-testdata.txt>     (run by tests/test_cpp_host_layer.py, -m gpu)
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <fstream>
#include <functional>
#include <random>
#include <sstream>
#include <thread>

#include "../../lorikeet_amd/csrc/host/lorikeet_pair_hmm.hpp"

using namespace lorikeet;

static int g_checks = 0;
#define ASSERT(cond, ...)                                                  \
    do {                                                                   \
        ++g_checks;                                                        \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "ASSERT FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
            std::fprintf(stderr, __VA_ARGS__);                             \
            std::fprintf(stderr, "\n");                                    \
            throw std::runtime_error("assertion failed");                  \
        }                                                                  \
    } while (0)

static bool relative_eq(double a, double b, double epsilon) {  // approx::relative_eq! with only epsilon given
    if (a == b) return true;
    const double d = std::fabs(a - b);
    if (d <= epsilon) return true;
    return d <= std::max(std::fabs(a), std::fabs(b)) * 2.220446049250313e-16;
}
static bool is_valid_log10_probability(double v) { return v <= 0.0; }  // MathUtils::is_valid_log10_probability

// ---------------------------------------------------------------------------------------------------------
// tests/vector_pair_hmm_unit_tests.rs:22-92
// ---------------------------------------------------------------------------------------------------------
static void test_likelihoods_avx(const std::string &path) {
    std::ifstream file(path);
    ASSERT(file.good(), "cannot open %s", path.c_str());
    std::string line;
    int n = 0;
    while (std::getline(file, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream tokens(line);
        std::string hap_s, bases_s, q_s, i_s, d_s, g_s;
        double expected_result;
        tokens >> hap_s >> bases_s >> q_s >> i_s >> d_s >> g_s >> expected_result;
        auto parse_qual = [](const std::string &s, uint8_t min) {
            Bytes q;
            for (char c : s) q.push_back(std::max<uint8_t>(min, (uint8_t)(c - 33)));
            return q;
        };
        const Bytes hap_bases = bytes(hap_s), read_bases = bytes(bases_s);
        const Bytes base_quals = parse_qual(q_s, 6), insertion_quals = parse_qual(i_s, 0), deletion_quals = parse_qual(d_s, 0),
                    gcp = parse_qual(g_s, 0);
        const double result = forward(hap_bases, read_bases, base_quals, insertion_quals, deletion_quals, gcp);
        ASSERT(std::fabs(result - expected_result) < 1e-5, "direct result %g expected %g", result, expected_result);

        Haplotype hap(hap_bases, true);
        HmmRead read(read_bases, base_quals);
        read.ins_quals = insertion_quals;
        read.del_quals = deletion_quals;
        std::map<size_t, std::vector<HmmRead>> read_map{{0, {read}}};
        std::vector<Haplotype> hap_vec{hap};
        PairHMM hmm = PairHMM::initialize(hap_vec, read_map, AVXMode::Hip);
        AlleleLikelihoods likelihoods(hap_vec, {0}, read_map);
        PairHMMInputScoreImputator score_imputator(gcp[0]);
        hmm.compute_log10_likelihoods(0, likelihoods, {read}, score_imputator);
        const auto &la = hmm.get_log_likelihood_array();
        ASSERT(std::fabs(la[0] - expected_result) < 1e-5,
               "Likelihood not in expected range for PairHMM implementation: got %g expected %g", la[0], expected_result);
        ASSERT(likelihoods.sample_matrix(0)(0, 0) == la[0], "scatter");
        ++n;
    }
    ASSERT(n == 104, "expected 104 vectors, read %d", n);
}

// ---------------------------------------------------------------------------------------------------------
// tests/pair_hmm_unit_tests.rs
// ---------------------------------------------------------------------------------------------------------
static const std::string CONTEXT = "ACGTAATGACGATTGCA";
static const std::string LEFT_FLANK = "GATTTATCATCGAGTCTGC";
static const std::string RIGHT_FLANK = "CATGGATCGTTATCAGCTATCTCGAGGGATTCACTTAACAGTTTTA";
static const uint8_t MASSIVE_QUAL = 100;
static const std::string BASES = "ACGT";

static std::string as_bytes(const std::string &b, bool left, bool right) {  // :156-165
    return (left ? LEFT_FLANK : "") + CONTEXT + b + CONTEXT + (right ? RIGHT_FLANK : "");
}

struct BasicLikelihoodTestProvider {  // :33-154
    std::string reference, read, ref_bases_with_context, read_bases_with_context;
    uint8_t base_qual, ins_qual, del_qual, gcp;
    size_t expected_qual;
    BasicLikelihoodTestProvider(const std::string &reference_, const std::string &read_, uint8_t bq, uint8_t iq, uint8_t dq,
                                size_t expected, uint8_t gcp_, bool left, bool right)
        : reference(reference_), read(read_), ref_bases_with_context(as_bytes(reference_, left, right)),
          read_bases_with_context(as_bytes(read_, false, false)), base_qual(bq), ins_qual(iq), del_qual(dq), gcp(gcp_),
          expected_qual(expected) {}
    double expected_log_likelihood() const {
        return (double)expected_qual / -10.0 + 0.03 + std::log10(1.0 / (double)ref_bases_with_context.size());
    }
    Bytes qual_as_bytes(uint8_t phred_qual, bool do_gop, bool anchor_indel) const {
        Bytes q(read_bases_with_context.size(), anchor_indel ? MASSIVE_QUAL : phred_qual);
        if (anchor_indel) {
            if (do_gop) q[CONTEXT.size()] = phred_qual;
            else for (size_t i = 0; i < read.size(); ++i) q[i + CONTEXT.size()] = phred_qual;
        }
        return q;
    }
    double calc_log10_likelihood(PairHMM &pair_hmm, bool anchor_indel) const {
        return pair_hmm.compute_read_likelihood_given_haplotype_log10(
            bytes(ref_bases_with_context), bytes(read_bases_with_context), qual_as_bytes(base_qual, false, anchor_indel),
            qual_as_bytes(ins_qual, true, anchor_indel), qual_as_bytes(del_qual, true, anchor_indel),
            qual_as_bytes(gcp, false, anchor_indel), true, std::nullopt);
    }
};

static void test_basic_likelihoods(PairHMM &hmm, const BasicLikelihoodTestProvider &cfg) {  // :300-326
    const double actual = cfg.calc_log10_likelihood(hmm, true), expected = cfg.expected_log_likelihood();
    ASSERT(relative_eq(actual, expected, 0.2), "Failed with hmm calc %g -> %g", actual, expected);
    ASSERT(is_valid_log10_probability(actual), "Bad log likelihood %g", actual);
}

static void make_basic_likelihood_tests() {  // :169-298
    PairHMM hmm = PairHMM::quick_initialize(0, 0);
    hmm.do_not_use_tristate_correction();
    int n = 0;
    for (uint8_t base_qual : {10, 20, 30, 40, 50})
        for (uint8_t indel_qual : {20, 30, 40, 50})
            for (uint8_t gcp : {8, 10, 20}) {
                for (char ref_base : BASES)
                    for (char read_base : BASES) {
                        test_basic_likelihoods(hmm, BasicLikelihoodTestProvider(std::string(1, ref_base), std::string(1, read_base),
                                                                                base_qual, indel_qual, indel_qual,
                                                                                ref_base == read_base ? 0 : base_qual, gcp, false, false));
                        ++n;
                    }
                for (int size : {2, 3, 4, 5, 7, 8, 9, 10, 20, 30, 35})
                    for (char base : BASES) {
                        const size_t expected = (size_t)(indel_qual + (size - 2) * gcp);
                        for (bool insertion_p : {true, false}) {
                            const std::string small(1, base), big((size_t)size, base);
                            const std::string reference = insertion_p ? small : big, read = insertion_p ? big : small;
                            for (auto lr : {std::pair<bool, bool>{false, false}, {true, false}, {false, true}, {true, true}}) {
                                test_basic_likelihoods(hmm, BasicLikelihoodTestProvider(reference, read, base_qual, indel_qual, indel_qual,
                                                                                        expected, gcp, lr.first, lr.second));
                                ++n;
                            }
                        }
                    }
            }
    ASSERT(n == 5 * 4 * 3 * (16 + 11 * 4 * 2 * 4), "case count %d", n);
}

static void mismatch_every_position(const std::string &hap_s, bool centred) {  // :328-405
    const Bytes haplotype_1 = bytes(hap_s);
    const uint8_t match_qual = 90, mismatch_qual = 20, indel_qual = 80;
    const size_t offset = 2, n = haplotype_1.size() - (centred ? 2 * offset : offset);
    const Bytes gop(n, indel_qual), gcp(n, indel_qual);
    PairHMM logless_hmm = PairHMM::quick_initialize(n, haplotype_1.size());
    logless_hmm.do_not_use_tristate_correction();
    for (size_t k = 0; k < n; ++k) {
        Bytes quals(n, match_qual);
        quals[k] = mismatch_qual;  // one base mismatches the haplotype
        Bytes m_read(haplotype_1.begin() + offset, centred ? haplotype_1.end() - offset : haplotype_1.end());
        m_read[k] = m_read[k] == 'C' ? 'T' : 'C';
        const double res_1 = logless_hmm.compute_read_likelihood_given_haplotype_log10(haplotype_1, m_read, quals, gop, gop, gcp,
                                                                                       true, std::nullopt);
        const double expected = std::log10((1.0 / (double)haplotype_1.size()) *
                                           std::pow(QualityUtils::qual_to_prob(match_qual), (double)(m_read.size() - 1)) *
                                           QualityUtils::qual_to_error_prob(mismatch_qual));
        ASSERT(relative_eq(res_1, expected, 1e-2), "Result %g, Expected %g", res_1, expected);
    }
}

static double get_expected_matching_log_likelihood(size_t read_len, size_t ref_len, uint8_t base_qual, uint8_t ins_qual) {  // :471-492
    const double ic = std::fabs((double)ref_len - (double)read_len + 1.0) / (double)ref_len;
    if (read_len < ref_len) return std::log10(ic * std::pow(QualityUtils::qual_to_prob(base_qual), (double)read_len));
    if (read_len > ref_len)
        return std::log10(ic * std::pow(QualityUtils::qual_to_prob(base_qual), (double)ref_len) *
                          std::pow(QualityUtils::qual_to_error_prob(ins_qual), (double)read_len - (double)ref_len));
    return 0.0;
}

static double run_flat(const Bytes &ref, const Bytes &read, uint8_t bq, uint8_t iq, uint8_t dq, uint8_t gcp) {
    PairHMM hmm = PairHMM::quick_initialize(read.size(), ref.size());
    hmm.do_not_use_tristate_correction();
    const size_t n = read.size();
    return hmm.compute_read_likelihood_given_haplotype_log10(ref, read, Bytes(n, bq), Bytes(n, iq), Bytes(n, dq), Bytes(n, gcp), true,
                                                             std::nullopt);
}

static void hmm_providers() {  // hmm_provider_simple :522-527, make_hmm_provider :529-539
    for (size_t read_size : {1, 2, 5, 10}) {
        const Bytes read_bases(read_size, 'A');
        ASSERT(run_flat(read_bases, read_bases, 20, 37, 37, 10) <= 0.0, "test_read_same_as_haplotype");
        for (size_t ref_size : {1, 2, 5, 10})
            if (ref_size > read_size) {
                ASSERT(run_flat(bytes("CC" + std::string(ref_size, 'A') + "GGA"), read_bases, 20, 37, 37, 10) <= 0.0,
                       "test_multiple_read_matches_in_haplotype");
                const double d = run_flat(Bytes(ref_size, 'A'), read_bases, 20, 100, 100, 100);
                const double expected = get_expected_matching_log_likelihood(read_size, ref_size, 20, 100);
                ASSERT(relative_eq(d, expected, 1e-3),
                       "Likelihoods should sum to just the error prob of the read but got Result %g, Expected %g", d, expected);
            }
    }
}

static void make_big_read_hmm_provider() {  // :541-597
    const std::string read_1 = "ACCAAGTAGTCACCGT", ref_1 = "ACCAAGTAGTCACCGTAACG";
    for (int n_read_copies : {1, 2, 10, 20, 50})
        for (int n_ref_copies : {1, 2, 10, 20, 100})
            if (n_ref_copies > n_read_copies) {
                std::string read, reference;
                for (int i = 0; i < n_read_copies; ++i) read += read_1;
                for (int i = 0; i < n_ref_copies; ++i) reference += ref_1;
                const double d = run_flat(bytes(reference), bytes(read), 30, 40, 40, 10);  // up to 800 x 2000
                ASSERT(d <= 0.0 && !std::isnan(d), "test_really_big_reads %g", d);
            }
}

static void test_likelihoods_from_haplotypes() {  // :637-683
    const size_t read_size = 10, ref_size = 20;
    const Bytes read_bases(read_size, 'A'), ref_bases(ref_size, 'A');
    const uint8_t base_qual = 20, ins_qual = MASSIVE_QUAL;
    Haplotype ref_h(ref_bases, true);
    HmmRead read(read_bases, Bytes(read_size, base_qual));  // no BI/BD tags: flat Q45, as in the reference's test
    std::vector<HmmRead> reads{read};
    std::map<size_t, std::vector<HmmRead>> sample_evidence_map{{0, reads}};
    PairHMMInputScoreImputator input_score_imputator(MASSIVE_QUAL);
    PairHMM hmm = PairHMM::quick_initialize(read_size, ref_size);
    hmm.do_not_use_tristate_correction();
    AlleleLikelihoods haplotype_mat({ref_h}, {0}, sample_evidence_map);
    hmm.compute_log10_likelihoods(0, haplotype_mat, {}, input_score_imputator);
    ASSERT(hmm.get_log_likelihood_array().size() == 0, "empty read list is a no-op");
    hmm.compute_log10_likelihoods(0, haplotype_mat, reads, input_score_imputator);
    const double expected = get_expected_matching_log_likelihood(read_size, ref_size, base_qual, ins_qual);
    const auto &la = hmm.get_log_likelihood_array();
    ASSERT(la.size() == 1, "one likelihood");
    ASSERT(relative_eq(la[0], expected, 1e-3), "got Result %g, Expected %g", la[0], expected);
}

static void make_haplotype_indexing_provider() {  // :725-814: cached == full recompute, 1e-9
    const std::string prefix = "AACCGGTTTTTGGGCCCAAACGTACGTACAGTTGGTCAACATCGATCAGGTTCCGGAGTAC";
    const std::string root_1 = "ACGTGTCAAACCGGGTT", root_2 = "ACGTGTCACACTGGGTT", root_3 = "ACGTGTCACTCCGCGTT";
    for (const std::string &read_full : {std::string("ACGTGTCACACTGGATT"), root_1, root_2, std::string("ACGTGTCACACTGGATTCGAT"),
                                         std::string("CCAGTAACGTGTCACACTGGATTCGAT")})
        for (size_t read_length = 10; read_length < read_full.size(); read_length += 3) {
            const Bytes read = bytes(read_full.substr(0, read_length));
            const size_t n = read.size();
            PairHMM hmm = PairHMM::quick_initialize(n, prefix.size() + root_1.size());
            hmm.do_not_use_tristate_correction();
            auto calc = [&](const std::string &hap, const std::optional<std::string> &next, bool recache) {
                return hmm.compute_read_likelihood_given_haplotype_log10(
                    bytes(hap), read, Bytes(n, 30), Bytes(n, 45), Bytes(n, 40), Bytes(n, 10), recache,
                    next ? std::optional<Bytes>(bytes(*next)) : std::nullopt);
            };
            for (int prefix_start = (int)prefix.size(); prefix_start >= 0; prefix_start -= 9) {
                const std::string my_prefix = prefix.substr((size_t)prefix_start);
                const std::string hap_1 = my_prefix + root_1, hap_2 = my_prefix + root_2, hap_3 = my_prefix + root_3;
                calc(hap_1, hap_2, true);
                const double actual_2 = calc(hap_2, hap_3, false);
                const double expected_2 = calc(hap_2, std::nullopt, true);
                ASSERT(relative_eq(actual_2, expected_2, 1e-9), "HMM caching calculation failed: expected %g got %g", expected_2, actual_2);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// tests/pair_hmm_likelihood_calculation_engine_unit_tests.rs:21-88
// ---------------------------------------------------------------------------------------------------------
static void test_compute_likelihoods() {
    PairHMMLikelihoodCalculationEngine lce(93, MathUtils::log_to_log10(QualityUtils::qual_to_error_prob_log10(45)),
                                           PCRErrorModel::Conservative, 16, false, 1.0, 0.02, true, false, true, AVXMode::Hip);
    std::map<size_t, std::vector<HmmRead>> per_sample_read_list;
    const size_t n = 10;
    HmmRead read1(Bytes(n, 'A'), Bytes(n, 30));  // create_artificial_read_default("test", 0, 0, 10, false)
    read1.name = "test";
    read1.mapq = 60;
    per_sample_read_list[0] = {read1};
    const std::vector<size_t> sample{0};
    Bytes ref_bases(n + 1, 'A');
    Haplotype hap1(ref_bases, true);
    AssemblyResultSet assembly_result_set(hap1);
    assembly_result_set.add_haplotype(hap1);
    Bytes bases_modified = ref_bases;
    bases_modified[5] = 'C';
    Haplotype hap2(bases_modified, false);
    assembly_result_set.add_haplotype(hap2);
    AlleleLikelihoods likes = lce.compute_read_likelihoods(assembly_result_set, sample, per_sample_read_list);
    ASSERT(likes.alleles().size() == 2, "alleles");
    ASSERT(likes.evidence_count() == 1, "evidence");
    const double v1 = likes.sample_matrix(0)(0, 0), v2 = likes.sample_matrix(0)(1, 0);
    ASSERT(v1 > v2, "Matching hap should have a higher likelihood %g -> %g", v1, v2);
    // SURVEY.md section 4 known answers for this fixture (derived from the reference formulas)
    ASSERT(std::fabs(v1 - -0.7446794209931795) < 1e-9 && std::fabs(v2 - -2.6990045895578127) < 1e-9, "known answers %.16g %.16g", v1, v2);
}

static void test_error_behaviour() {
    // pair_hmm.rs:425-440 asserts
    bool threw = false;
    try {
        forward(bytes("ACGT"), bytes("ACG"), Bytes(3, 30), Bytes(2, 40), Bytes(3, 40), Bytes(3, 10));
    } catch (const Panic &e) {
        threw = std::string(e.what()) == "Read bases and insertion gcp aren't the same size";
    }
    ASSERT(threw, "length mismatch must panic with the reference's message");
    // pair_hmm.rs:478-481: an improper model (Q0 gap-open) makes a matching read score > 0
    threw = false;
    try {
        forward(Bytes(20, 'A'), Bytes(12, 'A'), Bytes(12, 60), Bytes(12, 0), Bytes(12, 0), Bytes(12, 60));
    } catch (const Panic &e) {
        threw = std::string(e.what()) == "PairHmm Log Probability cannot be greater than 0.0";
    }
    ASSERT(threw, "positive result must panic with the reference's message");
    threw = false;
    try {
        pcr_error_model_from_arg("bogus");
    } catch (const Panic &e) {
        threw = std::string(e.what()) == "Unknown PCR Error Model";
    }
    ASSERT(threw, "Unknown PCR Error Model");
}

// ---------------------------------------------------------------------------------------------------------
// The reference's threading (assembly_region_walker.rs:210-273): rayon workers, each with a clone of the engine per
// task, one region per compute_read_likelihoods call, all at once.  Every worker must get what a lone caller gets.
// ---------------------------------------------------------------------------------------------------------
struct WorkerRegion {
    std::vector<Haplotype> haps;
    std::vector<HmmRead> reads;
};

static WorkerRegion random_region(uint64_t seed) {
    uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
    auto rnd = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return (uint32_t)(x >> 20);
    };
    const char acgt[] = "ACGT";
    WorkerRegion g;
    const size_t H = 80 + rnd() % 200, nh = 1 + rnd() % 5, nr = 1 + rnd() % 20;
    Bytes root(H);
    for (auto &b : root) b = acgt[rnd() & 3];
    for (size_t a = 0; a < nh; ++a) {
        Bytes h = root;
        if (a) h[rnd() % H] = acgt[rnd() & 3], h[rnd() % H] = acgt[rnd() & 3];
        Haplotype hap(h, a == 0);
        bool dup = false;
        for (const auto &o : g.haps) dup |= o.get_bases() == h;
        if (!dup) g.haps.push_back(hap);
    }
    for (size_t r = 0; r < nr; ++r) {
        const size_t n = 20 + rnd() % std::min<size_t>(100, H - 20), s = rnd() % (H - n + 1);
        Bytes b(root.begin() + s, root.begin() + s + n), q(n);
        for (size_t i = 0; i < n; ++i) {
            if (rnd() % 50 == 0) b[i] = acgt[rnd() & 3];
            q[i] = (uint8_t)(10 + rnd() % 30);
        }
        HmmRead read(b, q);
        read.mapq = (uint8_t)(rnd() % 3 ? 60 : 20);
        g.reads.push_back(read);
    }
    return g;
}

static std::vector<double> region_values(const WorkerRegion &g, bool dynamic) {
    PairHMMLikelihoodCalculationEngine lce(10, MathUtils::log_to_log10(QualityUtils::qual_to_error_prob_log10(45)),
                                           PCRErrorModel::Conservative, 18, dynamic, 1.0, 0.02, true, false, true, AVXMode::Hip);
    AssemblyResultSet ars(g.haps[0]);
    for (const auto &h : g.haps) ars.add_haplotype(h);
    std::map<size_t, std::vector<HmmRead>> per_sample{{0, g.reads}};
    AlleleLikelihoods likes = lce.compute_read_likelihoods(ars, {0}, per_sample);
    std::vector<double> v{(double)likes.evidence_count()};
    const Matrix &m = likes.sample_matrix(0);
    for (size_t a = 0; a < likes.alleles().size(); ++a)
        for (size_t r = 0; r < likes.evidence_count(); ++r) v.push_back(m(a, r));
    return v;
}

static void test_rayon_worker_pattern() {
    const int T = 8, per_thread = 24;
    std::vector<std::vector<WorkerRegion>> regions(T);
    std::vector<std::vector<std::vector<double>>> want(T);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < per_thread; ++k) {
            regions[t].push_back(random_region(1000 * t + k));
            want[t].push_back(region_values(regions[t].back(), (t + k) % 2));  // lone caller
        }
    std::atomic<int> mismatches{0}, panics{0}, done{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (int rep = 0; rep < 3; ++rep)
                for (int k = 0; k < per_thread; ++k) {
                    try {
                        const std::vector<double> got = region_values(regions[t][k], (t + k) % 2);
                        bool same = got.size() == want[t][k].size();
                        for (size_t i = 0; same && i < got.size(); ++i) same = std::fabs(got[i] - want[t][k][i]) <= 1e-12;
                        if (!same) ++mismatches;
                    } catch (const std::exception &) {
                        ++panics;
                    }
                    ++done;
                }
        });
    for (auto &x : th) x.join();
    ASSERT(done == T * per_thread * 3, "all calls returned");
    ASSERT(panics == 0, "%d calls panicked", panics.load());
    ASSERT(mismatches == 0, "%d regions differ from the lone caller's results", mismatches.load());
    // a worker whose region trips the reference's assert panics alone; its neighbours are served
    std::atomic<int> bad_panics{0}, good_ok{0};
    std::vector<std::thread> th2;
    for (int t = 0; t < 6; ++t)
        th2.emplace_back([&, t] {
            for (int k = 0; k < 20; ++k) {
                try {
                    if (t == 0) {
                        forward(Bytes(20, 'A'), Bytes(12, 'A'), Bytes(12, 60), Bytes(12, 0), Bytes(12, 0), Bytes(12, 60));
                    } else {
                        const double v = forward(bytes("ACGTACGTACGTACGTAAAC"), bytes("ACGTACGTACG"), Bytes(11, 30), Bytes(11, 40),
                                                 Bytes(11, 40), Bytes(11, 10));
                        if (v < 0.0 && v > -3.0) ++good_ok;
                    }
                } catch (const Panic &e) {
                    if (t == 0 && std::string(e.what()) == "PairHmm Log Probability cannot be greater than 0.0") ++bad_panics;
                }
            }
        });
    for (auto &x : th2) x.join();
    ASSERT(bad_panics == 20, "the faulty worker panicked %d of 20 times", bad_panics.load());
    ASSERT(good_ok == 100, "%d of 100 neighbouring calls were served", good_ok.load());
}

// The same pattern for the WHOLE per-region path (haplotype_caller_engine.rs:1311-1357: compute_read_likelihoods, then
// realign_reads_to_their_best_haplotype): every worker hands one region at a time to phmm_region_submit / phmm_wait on
// the one shared handle; whatever regions share a flush, every field a worker gets back equals what a lone caller of
// phmm_region_compute gets (likelihoods to 1e-12: which regions share a launch selects the kernel shape).
struct RegionIo {
    std::vector<uint32_t> rro, rho, ro, ho, hco, hc, hs, oco, oc, cig, n_cig;
    std::vector<uint64_t> oo, rstart, outco;
    Bytes bases, quals, mapq, haps, keep;
    std::vector<int32_t> ref, pri, best, status;
    std::vector<double> out, lk, conf;
    std::vector<int64_t> pos;
};
static RegionIo region_io(const WorkerRegion &g) {
    RegionIo io;
    io.rro = {0, (uint32_t)g.reads.size()};
    io.rho = {0, (uint32_t)g.haps.size()};
    io.ro = {0};
    io.ho = {0};
    io.hco = {0};
    io.oco = {0};
    io.outco = {0};
    for (const auto &r : g.reads) {
        io.bases.insert(io.bases.end(), r.bases.begin(), r.bases.end());
        io.quals.insert(io.quals.end(), r.quals.begin(), r.quals.end());
        io.mapq.push_back(r.mapq);
        io.ro.push_back((uint32_t)io.bases.size());
        io.oc.push_back((uint32_t)r.bases.size() << 4);  // <length>M
        io.oco.push_back((uint32_t)io.oc.size());
        io.outco.push_back(io.outco.back() + 12);
    }
    for (const auto &h : g.haps) {
        io.haps.insert(io.haps.end(), h.get_bases().begin(), h.get_bases().end());
        io.ho.push_back((uint32_t)io.haps.size());
        io.hc.push_back((uint32_t)h.get_bases().size() << 4);
        io.hco.push_back((uint32_t)io.hc.size());
        io.hs.push_back(0);
        io.pri.push_back(AssemblyBasedCallerUtils::haplotype_alignment_tiebreaking_priority(h));
    }
    const size_t nr = g.reads.size(), nh = g.haps.size();
    io.oo = {0, (uint64_t)nr * nh};
    io.rstart = {4000};
    io.ref = {0};
    io.out.assign(nr * nh, 0.0);
    io.keep.assign(nr, 0);
    io.best.assign(nr, 0);
    io.status.assign(nr, 0);
    io.lk.assign(nr, 0.0);
    io.conf.assign(nr, 0.0);
    io.pos.assign(nr, 0);
    io.cig.assign(12 * nr, 0);
    io.n_cig.assign(nr, 0);
    return io;
}
static int region_call(phmm_handle *h, RegionIo &io, bool dynamic, bool shared) {
    phmm_engine_config cfg{};
    cfg.constant_gcp = 10;
    cfg.pcr_error_model = 3;
    cfg.base_quality_score_threshold = 18;
    cfg.dynamic_read_disqualification = dynamic;
    cfg.symmetrically_normalize_alleles_to_reference = 1;
    cfg.log10_global_read_mismapping_rate = -4.5;
    cfg.read_disqualification_scale = 1.0;
    cfg.expected_error_rate_per_base = 0.02;
    const phmm_realign_config rcfg{{10, -15, -30, -5}, PHMM_SW_SOFTCLIP, PHMM_REGION_SKIP_SINGLE_ALLELE, 0.2};
    if (!shared)
        return phmm_region_compute(h, &cfg, &rcfg, 1, io.rro.data(), io.rho.data(), io.ro.data(), io.bases.data(), io.quals.data(), nullptr, nullptr,
                                   io.mapq.data(), nullptr, io.ho.data(), io.haps.data(), io.ref.data(), io.oo.data(), io.pri.data(), io.rstart.data(),
                                   io.hco.data(), io.hc.data(), io.hs.data(), io.oco.data(), io.oc.data(), io.outco.data(), io.out.data(), io.keep.data(),
                                   io.best.data(), io.lk.data(), io.conf.data(), io.cig.data(), io.n_cig.data(), io.pos.data(), io.status.data());
    uint64_t ticket = 0;
    const int rc = phmm_region_submit(h, &cfg, &rcfg, 1, io.rro.data(), io.rho.data(), io.ro.data(), io.bases.data(), io.quals.data(), nullptr, nullptr,
                                      io.mapq.data(), nullptr, io.ho.data(), io.haps.data(), io.ref.data(), io.oo.data(), io.pri.data(), io.rstart.data(),
                                      io.hco.data(), io.hc.data(), io.hs.data(), io.oco.data(), io.oc.data(), io.outco.data(), io.out.data(), io.keep.data(),
                                      io.best.data(), io.lk.data(), io.conf.data(), io.cig.data(), io.n_cig.data(), io.pos.data(), io.status.data(), &ticket);
    return rc ? rc : phmm_wait(h, ticket);
}
static void test_region_pipeline_worker_pattern() {
    const int T = 8, per_thread = 16;
    std::vector<std::vector<WorkerRegion>> regions(T);
    std::vector<std::vector<RegionIo>> want(T), got(T);
    detail::Handle lone = detail::make_handle(0, 0);
    int realigned = 0, single = 0;
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < per_thread; ++k) {
            regions[t].push_back(random_region(77000 + 100 * t + k));
            want[t].push_back(region_io(regions[t].back()));
            got[t].push_back(region_io(regions[t].back()));
            ASSERT(region_call(lone.get(), want[t].back(), (t + k) % 2, false) == PHMM_OK, "lone call: %s", phmm_last_error(lone.get()));
            for (int32_t st : want[t].back().status) realigned += st == PHMM_PROJECT_REALIGNED;
            single += regions[t].back().haps.size() == 1;
        }
    ASSERT(realigned > 200 && single > 0, "%d reads realigned, %d single-allele regions", realigned, single);
    phmm_handle *shared = detail::shared_handle(0);
    std::atomic<int> failed{0}, mismatches{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (int rep = 0; rep < 3; ++rep)
                for (int k = 0; k < per_thread; ++k) {
                    RegionIo &g = got[t][k];
                    const RegionIo &w = want[t][k];
                    if (region_call(shared, g, (t + k) % 2, true) != PHMM_OK) {
                        ++failed;
                        continue;
                    }
                    bool same = g.keep == w.keep && g.best == w.best && g.status == w.status && g.pos == w.pos && g.n_cig == w.n_cig;
                    for (size_t r = 0; same && r < g.n_cig.size(); ++r)  // (what lies behind a CIGAR's last element is not defined)
                        same = std::equal(g.cig.begin() + g.outco[r], g.cig.begin() + g.outco[r] + g.n_cig[r], w.cig.begin() + w.outco[r]);
                    for (size_t i = 0; same && i < g.out.size(); ++i) same = std::fabs(g.out[i] - w.out[i]) <= 1e-12;
                    auto close = [](double a, double b) { return a == b || (std::isnan(a) && std::isnan(b)) || std::fabs(a - b) <= 1e-12; };  // (-inf / NaN: no best allele)
                    for (size_t i = 0; same && i < g.lk.size(); ++i) same = close(g.lk[i], w.lk[i]) && close(g.conf[i], w.conf[i]);
                    if (!same) ++mismatches;
                }
        });
    for (auto &x : th) x.join();
    ASSERT(failed == 0, "%d calls failed: %s", failed.load(), phmm_last_error(shared));
    ASSERT(mismatches == 0, "%d regions differ from the lone caller's results", mismatches.load());
    uint64_t flushes = 0, subs = 0;
    phmm_submit_stats(shared, &flushes, &subs);
    ASSERT(flushes <= subs, "flushes %llu, submissions %llu", (unsigned long long)flushes, (unsigned long long)subs);
}

// ---------------------------------------------------------------------------------------------------------
// tests/smith_waterman_aligner_unit_tests.rs: the asserted cases (:228-318, :380-400) and the flank-length
// property (:320-378), through the mirrored SmithWatermanAligner
// ---------------------------------------------------------------------------------------------------------
static void assert_alignment_matches_expected(const std::string &reference, const std::string &read, int expected_start,
                                              const std::string &expected_cigar, const Parameters &weights,
                                              OverhangStrategy strategy) {  // :200-226
    const auto alignment = SmithWatermanAligner::align(bytes(reference), bytes(read), weights, strategy, detect_mode());
    ASSERT(alignment.get_alignment_offset() == expected_start, "offset %d, expected %d (%s)", alignment.get_alignment_offset(),
           expected_start, expected_cigar.c_str());
    ASSERT(alignment.get_cigar() == expected_cigar, "cigar %s, expected %s", alignment.get_cigar().c_str(), expected_cigar.c_str());
}

static void test_smith_waterman_asserted_cases() {
    assert_alignment_matches_expected("AAAGGACTGACTG", "ACTGACTGACTG", 1, "12M", ORIGINAL_DEFAULT, OverhangStrategy::SoftClip);   // :229-231
    assert_alignment_matches_expected("AAAGACTACTG", "AACGGACACTG", 1, "2M2I3M1D4M", Parameters{50, -100, -220, -12}, OverhangStrategy::SoftClip);  // :254-260
    assert_alignment_matches_expected("AAAGACTACTG", "AACGGACACTG", 0, "11M", Parameters{200, -50, -300, -22}, OverhangStrategy::SoftClip);        // :261-267
    const std::string matc = "CCCCC";
    assert_alignment_matches_expected("AAA" + matc, matc + "GGG", 3, "5M3S", ORIGINAL_DEFAULT, OverhangStrategy::SoftClip);         // :288-302
    assert_alignment_matches_expected("TGTGTGTGTGTGTGACAGAGAGAGAGAGAGAGAGAGAGAGAGAGA", "ACAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGA",
                                      14, "31M20S", STANDARD_NGS, OverhangStrategy::SoftClip);                                    // :305-318
    assert_alignment_matches_expected("AAA" + matc, matc, 3, "5M", ORIGINAL_DEFAULT, OverhangStrategy::SoftClip);                   // :381-386
    assert_alignment_matches_expected("AAA" + matc, matc, 0, "3D5M", ORIGINAL_DEFAULT, OverhangStrategy::InDel);
    assert_alignment_matches_expected("AAA" + matc, matc, 0, "3D5M", ORIGINAL_DEFAULT, OverhangStrategy::LeadingInDel);
    assert_alignment_matches_expected("AAA" + matc, matc, 3, "5M", ORIGINAL_DEFAULT, OverhangStrategy::Ignore);
    bool panicked = false;
    try {
        SmithWatermanAligner::align(bytes(""), bytes("ACGT"), ORIGINAL_DEFAULT, OverhangStrategy::SoftClip);
    } catch (const Panic &) {
        panicked = true;
    }
    ASSERT(panicked, "empty sequences must panic like the reference asserts");
}

static void test_for_identical_alignments_with_differing_flank_lengths() {  // :320-378
    auto strip = [](std::string s) {
        s.erase(std::remove(s.begin(), s.end(), '-'), s.end());
        return s;
    };
    const std::string core_ref = "CTTTAAGCCTGAGCCCCGCCCCCTGGCTCCCCGCCCCCTCTTCTCCCCTCCCCCAAGCCAGCACCTGGTGCCCCGGCGGGTCGTGCGGCGCGGCGCTCCGCGGTGAGCGCCTGACCCCGAGGGGGCCCGGGGCCGCGTCCCTGGGCCCTCCCCACCCTTGCGGTGGCCTCGCGGGTCCCAGGGGCGGGGCTGGAGCGGCAGCAGGGCCGGGGAGATGGGCGGTGGGGAGCGCGGGAGGGA";
    const std::string core_hap = strip("CTTTAAGCCTGAGCCCCGCCCCCTGGCTCCCCGCCCCCTCTTCTCCCCTCCCCCAAGCCAGCACCTGGTGCCCCGGCGGGTCGTGCGGCGCGGCGCTCCGCGGTGAGCGCCTGACCCCGA---------GGGCC--------GGGCCCTCCCCACCCTTGCGGTGGCCTCGCGGGTCCCAGGGGCGGGGCTGGAGCGGCAGCAGGGCCGGGGAGATGGGCGGTGGGGAGCGCGGGAGGGA");
    const std::string left = "GCGTCGCAGTCTTAAGGCCCCGCCTTTTCAGACAGCTTCCGCTGGGCCTGGGCCGCTGCGGGGCGGTCACGGCCC", right = "CCGGGCCGAGCCGGGGGAAGGGCTCCGGTGACT";
    const std::string pad = "NNNNNNNNNN";
    const auto flanked = SmithWatermanAligner::align(bytes(pad + left + core_ref + right + pad), bytes(pad + left + core_hap + right + pad),
                                                     NEW_SW_PARAMETERS, OverhangStrategy::SoftClip);
    const auto bare = SmithWatermanAligner::align(bytes(pad + core_ref + pad), bytes(pad + core_hap + pad), NEW_SW_PARAMETERS,
                                                  OverhangStrategy::SoftClip);
    // the indel elements of the two alignments are the same (type and length), only the flanking M differ
    std::vector<uint32_t> a, b;
    for (uint32_t e : flanked.cigar) if ((e & 15) != 0) a.push_back(e);
    for (uint32_t e : bare.cigar) if ((e & 15) != 0) b.push_back(e);
    ASSERT(flanked.cigar.size() == bare.cigar.size() && a == b && a.size() >= 2, "%s vs %s", flanked.get_cigar().c_str(),
           bare.get_cigar().c_str());
}

// tests/allele_likelihoods_unit_tests.rs:250-365 (test_best_alleles): random likelihoods, the reference allele has priority
// 1 and the others 0 -- the best allele is the arg-max unless the reference is within 0.2 of it, then the reference; its
// likelihood and confidence follow.  Then the realignment step on top (assembly_based_caller_utils.rs:208-246): reads cut
// out of their best haplotype align to it as one M element at the place they were cut from.
static void test_best_alleles() {
    std::mt19937_64 rng(20250928);
    std::normal_distribution<double> normal(0.0, 1.0);
    auto random_bases = [&](size_t n) {
        Bytes b(n);
        for (auto &c : b) c = "ACGT"[rng() & 3];
        return b;
    };
    for (size_t n_alleles : {1u, 2u, 3u, 5u, 8u})
        for (size_t ref_at : {size_t(0), n_alleles - 1}) {
            std::vector<Haplotype> alleles;
            for (size_t a = 0; a < n_alleles; ++a) alleles.emplace_back(random_bases(120 + 7 * a), a == ref_at);
            const std::vector<size_t> samples{0, 1, 2};
            std::map<size_t, std::vector<HmmRead>> reads;
            for (size_t s : samples)
                for (size_t r = 0; r < 5 + 9 * s; ++r) reads[s].push_back(HmmRead(random_bases(30), Bytes(30, 30)));
            AlleleLikelihoods original(alleles, samples, reads);
            for (size_t s : samples)
                for (size_t a = 0; a < n_alleles; ++a)
                    for (size_t r = 0; r < reads[s].size(); ++r) original.sample_matrix(s)(a, r) = normal(rng);
            const auto best_alleles = AssemblyBasedCallerUtils::best_alleles_breaking_ties_main(original, AssemblyBasedCallerUtils::reference_tiebreaking_priority);
            size_t seen = 0;
            for (const BestAllele &ba : best_alleles) {
                const Matrix &m = original.sample_matrix(ba.sample_index);
                const size_t r = ba.evidence_index;
                size_t best_index = 0;
                double best_lk = -INFINITY, second_lk = -INFINITY;
                for (size_t a = 0; a < n_alleles; ++a) {  // :290-303
                    const double lk = m(a, r);
                    if (lk > best_lk) {
                        second_lk = best_lk;
                        best_lk = lk;
                        best_index = a;
                    } else if (lk > second_lk) {
                        second_lk = lk;
                    }
                }
                const double ref_lk = m(ref_at, r);
                const bool ref_override = ref_at != best_index && best_lk - ref_lk < BestAllele::LOG_10_INFORMATIVE_THRESHOLD;  // :317-322
                ASSERT(ba.allele_index && *ba.allele_index == (ref_override ? ref_at : best_index), "best allele of read %zu", r);
                auto relative_eq = [](double x, double y) { return x == y || std::fabs(x - y) < 1e-12; };  // (a lone allele: +inf)
                ASSERT(relative_eq(ba.likelihood, ref_override ? ref_lk : best_lk), "likelihood of read %zu", r);
                ASSERT(relative_eq(ba.confidence, ref_override ? ref_lk - best_lk : best_lk - second_lk), "confidence of read %zu", r);
                ++seen;
            }
            ASSERT(seen == 5 + 14 + 23, "one BestAllele per unit of evidence, %zu", seen);
        }
    // realignment: every read is a piece of one haplotype and likes that haplotype best
    std::vector<Haplotype> alleles;
    for (size_t a = 0; a < 4; ++a) alleles.emplace_back(random_bases(200), a == 0);
    std::map<size_t, std::vector<HmmRead>> reads;
    std::vector<std::pair<size_t, size_t>> origin;
    for (size_t r = 0; r < 40; ++r) {
        const size_t a = r % 4, start = (size_t)(rng() % 120);
        reads[0].push_back(HmmRead(Bytes(alleles[a].bases_.begin() + start, alleles[a].bases_.begin() + start + 60), Bytes(60, 30)));
        origin.push_back({a, start});
    }
    AlleleLikelihoods lk(alleles, {0}, reads);
    for (size_t r = 0; r < 40; ++r)
        for (size_t a = 0; a < 4; ++a) lk.sample_matrix(0)(a, r) = a == origin[r].first ? -1.0 : -9.0;
    std::vector<SmithWatermanAlignmentResult> aligned;
    const auto best = AssemblyBasedCallerUtils::best_alleles_breaking_ties_main(lk, AssemblyBasedCallerUtils::haplotype_alignment_tiebreaking_priority, &aligned);
    for (size_t r = 0; r < 40; ++r) {
        ASSERT(best[r].allele_index && *best[r].allele_index == origin[r].first && best[r].is_informative(), "best haplotype of read %zu", r);
        ASSERT(aligned[r].get_cigar() == "60M" && aligned[r].alignment_offset == (int32_t)origin[r].second, "read %zu: %s @ %d", r,
               aligned[r].get_cigar().c_str(), aligned[r].alignment_offset);
    }
}

// tests/alignment_utils_unit_tests.rs:156-290 (make_read_aligned_to_ref_data): the read realigned through its haplotype
// lands where the test says, with the CIGAR the test says.
static void test_read_aligned_to_ref(const std::string &read, const Haplotype &hap, const Haplotype &ref_hap, size_t ref_start,
                                     int64_t expected_start, const std::string &expected_cigar) {
    const auto aligned = AlignmentUtils::create_read_aligned_to_ref(bytes(read), parse_cigar("10M"), hap, ref_hap, ref_start);
    ASSERT(aligned.realigned && aligned.pos == expected_start && cigar_to_string(aligned.cigar) == expected_cigar,
           "read %s: %s @ %lld, expected %s @ %lld", read.c_str(), cigar_to_string(aligned.cigar).c_str(), (long long)aligned.pos,
           expected_cigar.c_str(), (long long)expected_start);
}

static void make_read_aligned_to_ref_data() {
    const std::string hap_bases = "ACTGAAGGTTCC";
    Haplotype all_m(hap_bases, false);
    all_m.set_cigar(parse_cigar(std::to_string(hap_bases.size()) + "M"));
    const std::string all_m_cigar = std::to_string(hap_bases.size()) + "M";
    for (int i = -1; i < (int)hap_bases.size(); ++i) {
        std::string read = hap_bases;
        if (i != -1) read[i] = 'A';
        test_read_aligned_to_ref(read, all_m, all_m, 10, 10, all_m_cigar);
    }
    for (int pad = 1; pad < 10; ++pad) {
        test_read_aligned_to_ref(std::string(pad, 'N') + hap_bases, all_m, all_m, 10, 10, std::to_string(pad) + "I" + all_m_cigar);
        test_read_aligned_to_ref(hap_bases + std::string(pad, 'N'), all_m, all_m, 10, 10, all_m_cigar + std::to_string(pad) + "I");
    }
    for (size_t ref_start = 1; ref_start < 10; ++ref_start)
        for (size_t hap_start = ref_start; hap_start < 10 + ref_start; ++hap_start) {
            Haplotype hap(hap_bases, false);
            hap.set_cigar(all_m.cigar);
            hap.set_alignment_start_hap_wrt_ref(hap_start);
            test_read_aligned_to_ref(hap_bases, hap, all_m, ref_start, (int64_t)(ref_start + hap_start), all_m_cigar);
        }
    {
        const std::string reference = "GGGATCCTGCTACAAAGGTGAAACCCAGGAGAGTGTGGAGTCCAGAGTGTTGCCAGGACCCAGGCACAGGCATTAGTGCCCGTTGGAGAAAACAGGGGAATCCCGAAGAAATGGTGGGTCCTGGCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC";
        const std::string haplotype = "GGGATCCTGCTACAAAGGTGAAACCCAGGAGAGTGTGGAGTCCAGAGTGTTGCCAGGACCCAGGCACAGGCATTAGTGCCCGTTGGAGAAAACGGGAATCCCGAAGAAATGGTGGGTCCTGGCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC";
        Haplotype hap(haplotype, false), ref_hap(reference, true);
        hap.set_cigar(parse_cigar("93M2D92M"));
        hap.set_alignment_start_hap_wrt_ref(553);
        ref_hap.set_cigar(parse_cigar(std::to_string(reference.size()) + "M"));
        test_read_aligned_to_ref("CCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC", hap, ref_hap, 13011, 13011 + 553 + 123, "64M");
    }
}

// tests/cigar_utils_unit_tests.rs:21-283 (make_test_compute_cigar_data), a cross-section (all 114 cases run through the C ABI
// in tests/test_calculate_cigar_hip.py)
static void make_test_compute_cigar_data() {
    auto test_compute_cigar = [](const std::string &s1, const std::string &s2, const std::string &expected) {
        const auto c = CigarUtils::calculate_cigar(bytes(s1), bytes(s2), OverhangStrategy::InDel, NEW_SW_PARAMETERS);
        ASSERT(c && cigar_to_string(*c) == expected, "%s vs %s: %s, expected %s", s1.c_str(), s2.c_str(), c ? cigar_to_string(*c).c_str() : "None",
               expected.c_str());
    };
    test_compute_cigar("ATGGAGGGGC", "ATGGTGGGGC", "10M");
    test_compute_cigar("ATGGAGGGGC", "ATGGAAAATGGGGC", "5M4I5M");
    test_compute_cigar("ATGGAAAAAGGGGC", "ATGGTGGGGC", "4M4D6M");
    test_compute_cigar("ATGGAAAAAAAAAAGGGGC", "ATGGAAAATGGGGC", "4M5D10M");
    test_compute_cigar("NNNTGTGTGTGTGTGTGACAGAGAGAGAGAGAGAGAGAGAGAGAGAGANNN", "NNNACAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGANNN", "3M6I48M");
    test_compute_cigar("TCCCCCGGGT", "TAAACCCCCT", "1M3I5M3D1M");
    test_compute_cigar("G", "", "1D");
    test_compute_cigar("", "C", "1I");
    test_compute_cigar("AAAAACC", "CCGGGGGG", "5D2M6I");
    test_compute_cigar("GX", "X", "1D1M");
    test_compute_cigar("XAAAAACC", "XCCGGGGGG", "1M5D2M6I");
    test_compute_cigar("XG", "X", "1M1D");
    test_compute_cigar("XXXXXXXXXXXXXGXXXXXXXXXXXXX", "XXXXXXXXXXXXXXXXXXXXXXXXXX", "13M1D13M");
}

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s pairhmm-testdata.txt\n", argv[0]);
        return 2;
    }
    if (detect_mode() != AVXMode::Hip) {
        std::fprintf(stderr, "no HIP device: the host layer has no CPU fallback\n");
        return 3;
    }
    const std::string fixture = argv[1];
    const std::vector<std::pair<const char *, std::function<void()>>> tests = {
        {"test_likelihoods_avx", [&] { test_likelihoods_avx(fixture); }},
        {"make_basic_likelihood_tests", make_basic_likelihood_tests},
        {"test_mismatch_in_every_position_in_the_read_with_centred_haplotype",
         [] { mismatch_every_position("TTCTCTTCTGTTGTGGCTGGTTTTCTCTTCTGTTGTGGCTGGTTTTCTCTTCTGTTGTGGCTGGTT", true); }},
        {"test_mismatch_in_every_position_in_the_read", [] { mismatch_every_position("TTCTCTTCTGTTGTGGCTGGTT", false); }},
        {"hmm_provider_simple+make_hmm_provider", hmm_providers},
        {"make_big_read_hmm_provider", make_big_read_hmm_provider},
        {"test_likelihoods_from_haplotypes", test_likelihoods_from_haplotypes},
        {"make_haplotype_indexing_provider", make_haplotype_indexing_provider},
        {"test_compute_likelihoods", test_compute_likelihoods},
        {"error_behaviour", test_error_behaviour},
        {"rayon_worker_pattern (threads share one engine handle)", test_rayon_worker_pattern},
        {"region_pipeline_worker_pattern (likelihoods + realignment, threads share one engine handle)", test_region_pipeline_worker_pattern},
        {"smith_waterman_asserted_cases", test_smith_waterman_asserted_cases},
        {"test_for_identical_alignments_with_differing_flank_lengths", test_for_identical_alignments_with_differing_flank_lengths},
        {"test_best_alleles + realignment to the best haplotype", test_best_alleles},
        {"make_read_aligned_to_ref_data", make_read_aligned_to_ref_data},
        {"make_test_compute_cigar_data", make_test_compute_cigar_data},
    };
    int failed = 0;
    for (const auto &t : tests) {
        const int before = g_checks;
        try {
            t.second();
            std::printf("PASS %s (%d checks)\n", t.first, g_checks - before);
        } catch (const std::exception &e) {
            std::printf("FAIL %s: %s\n", t.first, e.what());
            ++failed;
        }
    }
    std::printf("%d tests, %d failed, %d checks\n", (int)tests.size(), failed, g_checks);
    return failed ? 1 : 0;
}
