// Quality -> probability tables (host copies).  See phmm_tables.cpp.
#pragma once
#include <vector>

namespace phmm {
const std::vector<double> &table_eps();             // [256] 10^(-q/10)
const std::vector<double> &table_eps_third();       // [256] eps/3
const std::vector<double> &table_match_to_match();  // [256*257/2] triangular
double initial_condition();                         // 2^1020
double initial_condition_log10();
// PCR indel error model cache, 101 entries (engine.rs:169-193); model 1 Hostile, 2 Aggressive, 3 Conservative
std::vector<unsigned char> pcr_error_model_cache(int model);
}  // namespace phmm
