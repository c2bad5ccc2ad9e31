// api_shared.h -- functions one part of the C ABI defines (api_filters.cpp) and another part calls (api_postcorrect.cpp)
#pragma once
#include "api_internal.h"

namespace bssm_api {
// parameter block and tables of a functor model (ssm_nlg / ssm_sde); params_only: stop after the parameter block
int functor_tables(const bssm_fn_model* mod, bssm::FilterTables* ft, std::vector<double>* fpar, int* m_out, int* k_out,
                   double* params_only = nullptr);
int to_fn_model(const bssm_fn_model* mod, bssm::FnModel* f, int max_states = 8);
int functor_psi_tables(const bssm_fn_model* mod, const double* is_mode, int max_iter, double conv_tol, bssm::FilterTables* ft,
                       std::vector<double>* fpar, std::vector<double>* amats, double* approx_loglik, bool* ok, int* m_out);
double sample_sd(const std::vector<double>& x);   // arma::stddev
}  // namespace bssm_api
