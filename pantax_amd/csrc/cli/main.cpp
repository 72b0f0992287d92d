// pantax-hip -- command-line front end of the pipeline seam.  Takes the profiling flags of the
// reference CLI (cli.rs:114-164, 222-240; defaults resolved as in main.rs:102-171) and calls
// pantax_hip_profile, i.e. it stands where `pantax ... --species --strain` calls profile::profile.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "../../../include/pantax_hip.h"
#include "rccl_comm.hpp"

static void usage() {
    fprintf(stderr,
            "usage: pantax-hip -db <db_dir> --gaf <gfa_mapped.gaf> [-T <work_dir>] [--species] [--strain]\n"
            "  --short-read | --long-read     (sets --fr default 0.3 / 0.5)\n"
            "  --fr F  --fc F(0.46)  -a F(1e-4)  --sr F(0.85)  --sd F(0.2)  --shift true|false\n"
            "  --min_cov N  --min_depth N  --sample N (default 500000)  --sample_test  --ds a,b,c  --smode 0|1  --no-filter\n"
            "  --solver gurobi|highs|cplex|cbc|glpk   whose second-solve semantics to reproduce (the LP optimum is the same; default gurobi)\n"
            "  --force  -R <reads_classification.tsv>  --range-file F  --species-len-file F  --reads-binning-file F\n"
            "  --image-cache 0|1|2  device-ready graph images <db>/species_graph_info/<otu>.hipdb: 1 = use, 2 = use and write\n"
            "  --filter-gaf  first replace the GAF by its best alignment per read (long reads; alignment.rs:171-175, gaf_filter.rs)\n"
            "  --filter-only <in.gaf> [<out.gaf>]   just write <stem>_filtered.gaf (or <out.gaf>) and exit\n"
            "  --gfa (read species_gfa/*.gfa instead of species_graph_info/*.bin)  --zip serialize|lz|zstd  --round (2-decimal output)  --device N\n"
            "  --ranks N --rank r   one process per GPU (start N of them; device = r unless --device): every rank tokenises 1/N of the\n"
            "                       GAF, reads travel to the owner of their species over RCCL, rank 0 writes the tables\n"
            "                       (defaults from WORLD_SIZE / RANK / LOCAL_RANK when set); --comm-id-file F (default <wd>/.pantax_hip_rccl_id),\n"
            "                       --comm-nonce N (the same for all ranks of one launch; default from $TORCHELASTIC_RUN_ID / $MASTER_PORT; the id file must also be fresh),\n"
            "                       --comm-timeout S (default 300: a rank whose peers do not complete the RCCL bootstrap in time aborts it and exits non-zero)\n");
}

int main(int argc, char **argv) {
    pantax_hip_profiling_config c;
    memset(&c, 0, sizeof(c));
    std::string wd = "pantax_db_tmp";
    c.min_species_abundance = 1e-4; c.unique_trio_nodes_fraction = -1; c.unique_trio_nodes_mean_count_f = 0.46;
    c.single_cov_ratio = 0.85; c.single_cov_diff = 0.2; c.filtered = 1; c.full = 1; c.mode = 2; c.sample_nodes = 500000;
    c.zip = "serialize"; c.world_size = 1;
    bool long_read = false, filter_gaf = false;
    const char *filter_in = nullptr, *filter_out = nullptr;
    int device = -1, ranks = 0, rank = -1;
    std::string id_file;
    // per-launch nonce of the id file (rccl_comm.hpp): something that changes from one launch to the next where the launcher offers it
    // (torchrun: the run id + the restart count), else the rendezvous port; the file's age is checked in every case
    // (the launcher's variables are read here, once, before any thread exists: the one place the CLI looks at the environment)
    auto env = [](const char *name) -> const char * { return std::getenv(name); };
    uint64_t nonce = env("MASTER_PORT") ? strtoull(env("MASTER_PORT"), nullptr, 10) : 0;
    double comm_timeout = 300.0;     // --comm-timeout: seconds the whole RCCL bootstrap may take before this rank gives up and exits non-zero
    if (const char *rid = env("TORCHELASTIC_RUN_ID")) {
        uint64_t h = 0xcbf29ce484222325ull;
        for (const char *q = rid; *q; ++q) { h ^= (uint64_t)(unsigned char)*q; h *= 0x100000001b3ull; }
        if (const char *rc = env("TORCHELASTIC_RESTART_COUNT")) h = h * 31 + strtoull(rc, nullptr, 10);
        nonce ^= h ? h : 1;
    }
    if (const char *ev = env("WORLD_SIZE")) ranks = atoi(ev);
    if (const char *ev = env("RANK")) rank = atoi(ev);
    if (const char *ev = env("LOCAL_RANK")) device = atoi(ev);
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "-db" || a == "--db") c.db = next();
        else if (a == "--gaf") c.input_aln_file = next();
        else if (a == "-T") wd = next();
        else if (a == "--species" || a == "-s") c.species = 1;
        else if (a == "--strain" || a == "-S") c.strain = 1;
        else if (a == "--short-read") long_read = false;
        else if (a == "--long-read") long_read = true;
        else if (a == "--fr") c.unique_trio_nodes_fraction = atof(next());
        else if (a == "--fc") c.unique_trio_nodes_mean_count_f = atof(next());
        else if (a == "-a") c.min_species_abundance = atof(next());
        else if (a == "--sr") c.single_cov_ratio = atof(next());
        else if (a == "--sd") c.single_cov_diff = atof(next());
        else if (a == "--shift") c.shift = !strcasecmp(next(), "true");
        else if (a == "--min_cov") c.min_cov = atoll(next());
        else if (a == "--min_depth") c.min_depth = atoll(next());
        else if (a == "--sample") c.sample_nodes = atoi(next());
        else if (a == "--sample_test" || a == "--sample-test") c.sample_test = 1;      // cli.rs:230-232
        else if (a == "--solver") {                                                    // cli.rs:158-160: whose handling of the second solve is reproduced
            const char *sv = next();
            c.solver_semantics = !strcasecmp(sv, "highs") ? PANTAX_HIP_SEMANTICS_HIGHS : PANTAX_HIP_SEMANTICS_GUROBI;   // gurobi, cplex, cbc, glpk agree (profile.rs:1500, :1906, :2125, :2672)
        }
        else if (a == "--ds") c.designated_species = next();
        else if (a == "--smode") c.mode = atoi(next());
        else if (a == "--no-filter") c.filtered = 0;
        else if (a == "--force") c.force = 1;
        else if (a == "-R" || a == "--report") c.out_binning_file = next();
        else if (a == "--range-file") c.range_file = next();
        else if (a == "--species-len-file") c.species_len_file = next();
        else if (a == "--reads-binning-file") c.reads_binning_file = next();
        else if (a == "--gfa") c.zip = nullptr;
        else if (a == "--zip") c.zip = next();        // serialize | lz | zstd (main.rs: --zip)
        else if (a == "--round") c.full = 0;
        else if (a == "--filter-gaf") filter_gaf = true;
        else if (a == "--image-cache") c.image_cache = atoi(next());
        else if (a == "--filter-only") { filter_in = next(); if (i + 1 < argc && argv[i + 1][0] != '-') filter_out = argv[++i]; }
        else if (a == "--device") device = atoi(next());
        else if (a == "--ranks") ranks = atoi(next());
        else if (a == "--rank") rank = atoi(next());
        else if (a == "--comm-id-file") id_file = next();
        else if (a == "--comm-nonce") nonce = strtoull(next(), nullptr, 10);
        else if (a == "--comm-timeout") comm_timeout = atof(next());
        else { usage(); return 2; }
    }
    const bool use_rccl = ranks >= 1 && rank >= 0;   // also a one-rank world goes through the communicator when asked for
    if (use_rccl && rank >= ranks) { fprintf(stderr, "pantax-hip: --rank %d of --ranks %d\n", rank, ranks); return 2; }
    if (device < 0) device = use_rccl ? rank : 0;
    if (filter_in) {
        pantax_hip_ctx *fctx = nullptr;
        if (pantax_hip_init(&fctx, &device, 1) != 0) { fprintf(stderr, "pantax-hip: %s\n", pantax_hip_last_error(nullptr)); return 1; }
        uint64_t nl = 0, nr = 0, nw = 0;
        const int frc = pantax_hip_gaf_filter(fctx, filter_in, filter_out, &nl, &nr, &nw);
        if (frc != 0) fprintf(stderr, "pantax-hip: error %d: %s\n", frc, pantax_hip_last_error(fctx));
        else printf("Filtered GAF: %llu lines, %llu alignment records, %llu written\n", (unsigned long long)nl, (unsigned long long)nr, (unsigned long long)nw);
        pantax_hip_destroy(fctx);
        return frc == 0 ? 0 : 1;
    }
    if (!c.db || !c.input_aln_file) { usage(); return 2; }
    if (c.unique_trio_nodes_fraction < 0) c.unique_trio_nodes_fraction = long_read ? 0.5 : 0.3;   // main.rs:108-114
    c.wd = wd.c_str(); c.output_dir = wd.c_str();
    pantax_hip_ctx *ctx = nullptr;
    int rc = pantax_hip_init(&ctx, &device, 1);
    if (rc != 0) { fprintf(stderr, "pantax-hip: %s\n", pantax_hip_last_error(nullptr)); return 1; }
    // the communicator first: with --filter-gaf only rank 0 rewrites the shared GAF, and the others must not open it before
    // the rename (the sharded ingest assumes ONE immutable file) -- the all-reduce below is that barrier and carries a failure
    // of the filter to every rank, so all of them leave together
    RcclComm comm;
    if (use_rccl) {
        if (id_file.empty()) id_file = wd + "/.pantax_hip_rccl_id";
        if (!comm.init(rank, ranks, id_file, nonce, comm_timeout)) {
            fprintf(stderr, "pantax-hip: rank %d: %s\n", rank, comm.err.c_str());
            if (comm.abandoned) { if (rank == 0) ::unlink(id_file.c_str()); fflush(stderr); _exit(3); }   // the bootstrap thread is still inside RCCL: no teardown may wait for it
            pantax_hip_destroy(ctx);
            return 1;
        }
        c.rank = rank; c.world_size = ranks; c.comm_user = &comm;
        c.allreduce_sum = &RcclComm::allreduce_sum; c.alltoallv = &RcclComm::alltoallv; c.comm_device_buffers = 1;
    }
    if (filter_gaf) {   // alignment.rs:171-175: filter, then the filtered file takes the GAF's place
        int frc = 0;
        if (!use_rccl || rank == 0) {
            const std::string filtered = std::string(c.input_aln_file) + ".best.tmp";
            frc = pantax_hip_gaf_filter(ctx, c.input_aln_file, filtered.c_str(), nullptr, nullptr, nullptr);
            if (frc != 0) fprintf(stderr, "pantax-hip: error %d: %s\n", frc, pantax_hip_last_error(ctx));
            else if (rename(filtered.c_str(), c.input_aln_file) != 0) { fprintf(stderr, "pantax-hip: cannot replace %s\n", c.input_aln_file); frc = 1; }
        }
        double failed = frc != 0 ? 1.0 : 0.0;
        if (use_rccl && RcclComm::allreduce_sum(&comm, &failed, 1) != 0) { fprintf(stderr, "pantax-hip: rank %d: the exchange behind --filter-gaf failed\n", rank); failed = 1.0; }
        if (failed != 0.0) { if (use_rccl) comm.destroy(id_file); pantax_hip_destroy(ctx); return 1; }
    }
    rc = pantax_hip_profile(ctx, &c);
    if (rc != 0) fprintf(stderr, "pantax-hip: %serror %d: %s\n", use_rccl ? ("rank " + std::to_string(rank) + ": ").c_str() : "", rc, pantax_hip_last_error(ctx));
    if (use_rccl) comm.destroy(id_file);
    pantax_hip_destroy(ctx);
    return rc == 0 ? 0 : 1;
}
