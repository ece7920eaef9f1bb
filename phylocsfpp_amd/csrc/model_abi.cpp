// model_abi.cpp -- model files behind the C-ABI: pcsf_model_open and friends (include/phylocsf_amd.h).
// Wraps the host driver's reader (phylocsfpp_amd/host/model_io.hpp: ECM files, newick tree in newick_flatten
// numbering, species names, --species reduction, --mapping) so that every binding -- the C++ CLI, the Python
// package, a maintainer's own caller -- gets the arrays pcsf_create takes from ONE implementation.
// Reference: load_model, src/models.hpp:1757-1857.  No device code here; works without a GPU.
#include <cstring>
#include <string>

#include "../../include/phylocsf_amd.h"
#include "../host/model_io.hpp"

namespace pcsf {
void set_global_error(const std::string& msg);  // api.cpp: what pcsf_last_error(NULL) returns
}

struct pcsf_model {
    pcsf_host::Model m;
};

extern "C" {

int pcsf_model_open(pcsf_model** out, const char* prefix, const char* alias_file, const char* species_csv, const char* mapping_file)
{
    if (!out || !prefix) {
        pcsf::set_global_error("pcsf_model_open: null argument");
        return PCSF_ERR_INVALID_ARG;
    }
    *out = nullptr;
    try {
        std::string aliases = alias_file ? alias_file : "";
        if (!alias_file) {  // convention of the CLI: species_aliases.tsv next to the model files
            const std::string p = prefix;
            const size_t slash = p.find_last_of('/');
            const std::string cand = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/species_aliases.tsv";
            if (FILE* f = fopen(cand.c_str(), "r")) {
                fclose(f);
                aliases = cand;
            }
        }
        pcsf_model* h = new pcsf_model;
        try {
            h->m = pcsf_host::load_model(prefix, aliases, species_csv ? species_csv : "", mapping_file ? mapping_file : "");
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
    } catch (const std::exception& ex) {
        pcsf::set_global_error(std::string("pcsf_model_open: ") + ex.what());
        return PCSF_ERR_INVALID_ARG;
    }
    return PCSF_OK;
}

void pcsf_model_close(pcsf_model* m) { delete m; }

int pcsf_model_describe(const pcsf_model* m, pcsf_model_desc* desc)
{
    if (!m || !desc) return PCSF_ERR_INVALID_ARG;
    desc->n_nodes = m->m.n_nodes;
    desc->child1 = m->m.child1.data();
    desc->child2 = m->m.child2.data();
    desc->branch_len = m->m.branch_len.data();
    desc->S_coding = m->m.S_c.data();
    desc->pi_coding = m->m.pi_c.data();
    desc->S_noncoding = m->m.S_nc.data();
    desc->pi_noncoding = m->m.pi_nc.data();
    return PCSF_OK;
}

const double* pcsf_model_branch_len_double(const pcsf_model* m) { return m ? m->m.branch_len_d.data() : nullptr; }

const int16_t* pcsf_model_parents(const pcsf_model* m) { return m ? m->m.parent.data() : nullptr; }

const char* pcsf_model_label(const pcsf_model* m, int32_t node)
{
    if (!m || node < 0 || node >= m->m.n_nodes) return nullptr;
    return m->m.label[static_cast<size_t>(node)].c_str();
}

int32_t pcsf_model_leaf_of(const pcsf_model* m, const char* species_name)
{
    if (!m || !species_name) return -1;
    const auto it = m->m.name_to_leaf.find(pcsf_host::lower(species_name));
    return it == m->m.name_to_leaf.end() ? -1 : it->second;
}

}  // extern "C"
