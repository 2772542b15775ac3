/*
 * main.cpp — Eb/N0 sweep driver, mirroring reference main.cpp:17-231 on top of the GPU decode path.
 *
 * Reads ./Profile.txt (same file as the reference), sweeps StartSNR..EndSNR in SNRPass steps, stops a point
 * when TestFrame >= 1000 and ErrorFrame >= 20 (reference main.cpp:164, :209) and appends the same
 * Result.txt / Temp.txt rows (columns of main.cpp:113-116, :219-222).
 *
 * Parallelism: the reference starts one pinned pthread per hardware thread, each with seed[index]
 * (main.cpp:31-34, :166-172).  Here `--streams T` such workers (default 64) are partitioned into contiguous
 * ranges over `--gpus G` GPUs; one host thread per GPU drives its CSimulate, and the per-GPU counters are
 * summed on the host after each round exactly as reference main.cpp:174-182 does.  With T equal to the
 * reference's thread count the printed counters are identical to the reference's.
 *
 * One process per GPU: `--ranks N --rank r --comm-file F [--device d]` runs rank r of N cooperating processes.  The T streams
 * are cut into N contiguous ranges exactly like the G ranges above, every rank decodes its own range on its own GPU, and the
 * four counters are summed over RCCL after each round (lnsfaid_allreduce_counters: one 32-byte all-reduce), so every rank takes
 * the same stop decisions; rank 0 writes Result.txt.  The RCCL id travels through the file F (CommFile.h: rank 0 clears F,
 * publishes the id under `--run-id ID`, removes F once every rank has joined; the others accept only a fresh record of the
 * same run id and give up after `--comm-timeout S` seconds, exit code 3).
 */
#include <sys/time.h>

#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <thread>
#include <chrono>
#include <vector>

#include "CSimulate.h"
#include "CommFile.h"

using namespace std;

int main(int argc, char** argv)
{
    /* OpenMP workers must sleep, not spin, between the channel loops: a spinning pool starves the HIP runtime's
     * threads while the GPU decodes (set before the OpenMP runtime starts) */
    setenv("OMP_WAIT_POLICY", "passive", 0);
    /* a container usually sees every core of the host but owns a share of them: an OpenMP team as large as the machine
     * turns every parallel region into time-slicing.  16 workers unless OMP_NUM_THREADS says otherwise. */
    setenv("OMP_NUM_THREADS", "16", 0);
    int streams = 64, gpus = 1, max_rounds = 0, ranks = 1, rank = 0, device = -1;
    const char* comm_file = nullptr;
    const char* run_id = "";
    int comm_timeout_s = 120;
    const int64_t started_at = (int64_t)time(nullptr);
    bool device_frontend = false, force_collect = false, encode = false;
    const char* profile = "Profile.txt";
    const char* resume = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--streams") && i + 1 < argc) streams = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--profile") && i + 1 < argc) profile = argv[++i];
        else if (!strcmp(argv[i], "--dump-fixinput") && i + 1 < argc) g_dump_fixinput = argv[++i]; /* test hook */
        else if (!strcmp(argv[i], "--ranks") && i + 1 < argc) ranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank") && i + 1 < argc) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--comm-file") && i + 1 < argc) comm_file = argv[++i];
        else if (!strcmp(argv[i], "--run-id") && i + 1 < argc) run_id = argv[++i]; /* the same on every rank of a run, new for every run */
        else if (!strcmp(argv[i], "--comm-timeout") && i + 1 < argc) comm_timeout_s = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-rounds") && i + 1 < argc) max_rounds = atoi(argv[++i]); /* 0 = reference stop rule only */
        else if (!strcmp(argv[i], "--device-frontend")) device_frontend = true; /* channel + demapper + quantiser on the GPU */
        else if (!strcmp(argv[i], "--resume") && i + 1 < argc) resume = argv[++i]; /* lastSeed table of a Temp.txt (reference CONTINUE_SEED 1) */
        else if (!strcmp(argv[i], "--encode")) encode = true; /* random information bits + the encoder derived from H (reference FAKE_ENCODE 0) */
        else if (!strcmp(argv[i], "--collect")) force_collect = true; /* collectflag = 1 from the first call (reference: once FER < 1e-5) */
        else { fprintf(stderr, "usage: %s [--streams T] [--gpus G] [--profile Profile.txt] [--max-rounds R] [--device-frontend] [--encode] [--collect] [--resume Temp.txt] [--ranks N --rank r --comm-file F [--run-id ID] [--comm-timeout S] [--device d]]\n", argv[0]); return 2; }
    }
    if (streams < 1 || gpus < 1 || gpus > streams) { fprintf(stderr, "need 1 <= gpus <= streams\n"); return 2; }
    if (ranks < 1 || rank < 0 || rank >= ranks || (ranks > 1 && (gpus != 1 || !comm_file || ranks > streams))) {
        fprintf(stderr, "--ranks N needs --rank r in [0, N), --comm-file F, N <= streams, and one GPU per process\n");
        return 2;
    }
    if (device < 0) device = ranks > 1 ? rank : 0;
    const bool multi = ranks > 1 || comm_file != nullptr;

    Parameter_Simulation p_simulation;
    if (!ReadProfile(&p_simulation, profile)) {
        cerr << "Cannot open Profile\n"; /* reference CTool.cpp:591-596 */
        exit(EXIT_FAILURE);
    }

    /* contiguous stream ranges per GPU; never split a group (a stream IS a sequence of whole groups) */
    vector<CSimulate> simulate(gpus);
    for (int g = 0; g < gpus; ++g) {
        /* this process's share of the streams (all of them unless --ranks), then that share over its GPUs */
        const int r_first = (int)((long)streams * rank / ranks), r_last = (int)((long)streams * (rank + 1) / ranks);
        const int mine = r_last - r_first;
        const int first = r_first + (int)((long)mine * g / gpus), last = r_first + (int)((long)mine * (g + 1) / gpus);
        simulate[g].device_frontend = device_frontend;
        simulate[g].encode = encode;
        simulate[g].Initial(p_simulation, first, last - first, multi ? device : g);
    }
    if (multi) { /* RCCL communicator of this run: the id travels through --comm-file */
        uint8_t id[LNSFAID_COMM_ID_BYTES];
        static_assert(LNSFAID_COMM_ID_BYTES == COMMFILE_ID_BYTES, "CommFile.h carries one RCCL id");
        if (rank == 0) {
            commfile_clear(comm_file); /* nothing of an earlier run may be found by a rank that starts after this point */
            if (lnsfaid_comm_unique_id(id)) { cerr << "RCCL is not available\n"; exit(EXIT_FAILURE); }
            if (!commfile_publish(comm_file, run_id, id)) { cerr << "cannot write " << comm_file << "\n"; exit(EXIT_FAILURE); }
        } else if (commfile_fetch(comm_file, run_id, id, comm_timeout_s * 1000, started_at)) {
            cerr << "no RCCL id of run '" << run_id << "' in " << comm_file << " after " << comm_timeout_s << " s\n";
            return 3;
        }
        simulate[0].ldpc->CommInit(p_simulation.decode_method, ranks, rank, id);
        if (rank == 0) commfile_clear(comm_file); /* every rank has joined: the id must not outlive the run */
    }

    if (resume) {
        /* reference CChannel.cpp:4-41, :116-119: the generator states main wrote to Temp.txt ("{IX,IY,IZ}," one row per
         * worker, main.cpp:200-207) replace the seed table; the counters start again, as in the reference */
        ifstream tin(resume);
        if (!tin.is_open()) { cerr << "Cannot open " << resume << "\n"; exit(EXIT_FAILURE); }
        vector<array<unsigned long, 3>> rows;
        string line;
        while (getline(tin, line)) {
            unsigned long a, b, c;
            const size_t brace = line.find('{');
            if (brace != string::npos && sscanf(line.c_str() + brace, "{%lu,%lu,%lu}", &a, &b, &c) == 3) rows.push_back({ a, b, c });
        }
        if (ranks > 1) { cerr << "--resume is per process: run it with --ranks 1\n"; exit(EXIT_FAILURE); }
        if ((int)rows.size() < streams) { cerr << resume << " holds " << rows.size() << " generator states, need " << streams << "\n"; exit(EXIT_FAILURE); }
        int idx = 0;
        for (auto& s : simulate)
            for (auto& ch : s.channel) { ch.RS.IX = rows[idx][0]; ch.RS.IY = rows[idx][1]; ch.RS.IZ = rows[idx][2]; ++idx; }
    }

    ofstream fout(rank == 0 ? "Result.txt" : "/dev/null", std::ios::app); /* rank 0 reports */
    if (!fout.is_open()) { cerr << "Cannot open Result.txt\n"; exit(EXIT_FAILURE); }
    fout << endl
         << "********************************************************************************************************************************************" << endl;
    fout << "DATE:" << __DATE__ << endl << "Time" << __TIME__ << endl;
    fout << "codeFile:" << MATRIX_FILE << endl;
    fout << "DecodeMethod:" << p_simulation.decode_method << endl;
    fout << "MaxItertion:" << p_simulation.Max_Iteration << endl;
    fout << "Modulation Type" << p_simulation.mod_type << endl;
    fout << "InterLeave ModType" << p_simulation.interleavemod_type << endl;
    fout << "scale=" << p_simulation.scale << endl;
    fout << "factor_1=" << p_simulation.Factor_1 << endl << "factor_2=" << p_simulation.Factor_2 << endl;
    fout << "Punctue Number: " << _PunctureBits << " Shorten Bits" << _ShortenBits << " RATE: " << simulate[0].ldpc->m_Rate << endl;
    fout << "streams=" << streams << " gpus=" << gpus << " ranks=" << ranks << endl;
    fout << setw(5) << "Eb_N0" << '\t' << setw(20) << "TestFrame" << '\t' << setw(15) << "ErrorFrame" << '\t' << setw(20) << "ErrorBits"
         << '\t' << setw(20) << "FER" << '\t' << setw(20) << "BER" << '\t' << setw(15) << "LT3ErrBitFrame" << '\t' << setw(15) << "Time(s)" << '\t' << endl;
    fout.close();
    if (rank == 0) { /* reference main.cpp:75-83 */
        ofstream demodout("demod.txt", std::ios::app);
        if (!demodout.is_open()) { cerr << "Cannot open demod.txt\n"; exit(EXIT_FAILURE); }
        demodout << setw(5) << " Eb/N0" << '\t' << setw(20) << " ModFER" << '\t' << setw(20) << "ModBER" << '\t' << setw(20) << "ModSER" << '\t' << endl;
    }
    if (rank == 0) cout << setw(5) << "Eb_N0" << setw(20) << "TestFrame" << setw(15) << "ErrorFrame" << setw(20) << "ErrorBits" << setw(20) << "FER"
         << setw(20) << "BER" << setw(15) << "LT3ErrBitFrame" << setw(15) << "Time(s)" << setw(18) << "decode info Gb/s" << endl;

    for (float snr = p_simulation.snr_start; snr < p_simulation.snr_end; snr += p_simulation.snr_pass) {
        unsigned long TestFrame = 0, ErrorFrame = 0, ErrorBits = 0, LT3ErrBitFrame = 0;
        double BER = 0, FER = 0, decode_s = 0;
        for (const char* name : { "iterCount.txt", "errorindex.txt", "errorfloat.txt", "errordecode.txt" }) { /* reference main.cpp:145-157 */
            ofstream f(name, std::ios::app);
            f << "Eb/N0: " << setw(5) << snr << "scale=" << p_simulation.scale << endl;
        }
        if (force_collect) collectflag = 1;
        for (auto& s : simulate) { s.Configure(snr, p_simulation.decode_method); s.decode_seconds = 0; }
        timeval t_start, t_end;
        gettimeofday(&t_start, NULL);
        int rounds = 0;
        while (TestFrame < 1000 || ErrorFrame < 20) {
            vector<thread> workers; /* Start()/End() of the reference: create + join per round (CSimulate.cpp:255-278) */
            for (int g = 0; g < gpus; ++g) workers.emplace_back([&simulate, g]() { simulate[g].Run(); });
            for (auto& w : workers) w.join();
            unsigned long add[4] = { 0, 0, 0, 0 };
            for (auto& s : simulate) { /* reference main.cpp:174-182: the reference adds the running totals again each round */
                add[0] += s.TestFrame; add[1] += s.ErrorFrame; add[2] += s.ErrorBits; add[3] += s.LT3ErrBitFrame;
            }
            if (multi) simulate[0].ldpc->AllReduceCounters(add); /* the same sum over the ranks' GPUs */
            TestFrame += add[0]; ErrorFrame += add[1]; ErrorBits += add[2]; LT3ErrBitFrame += add[3];
            BER = (double)(ErrorBits > 0 ? ErrorBits : 1) / ((double)TestFrame * (NmoinsK - _ShortenBits));
            FER = (double)(ErrorFrame > 0 ? ErrorFrame : 1) / TestFrame;
            if (FER < 1E-5 || force_collect) collectflag = 1; /* reference main.cpp:190-192: dump the error frames from here on */
            ofstream tout(ranks > 1 ? ("Temp.txt.rank" + to_string(rank)).c_str() : "Temp.txt", std::ios::out); /* per rank: its own generator states */
            tout << setw(5) << snr << '\t' << setw(20) << TestFrame << '\t' << setw(15) << ErrorFrame << '\t' << setw(20) << ErrorBits << '\t'
                 << setw(20) << FER << '\t' << setw(20) << BER << '\t' << setw(15) << LT3ErrBitFrame << '\t' << endl;
            tout << "const unsigned long lastSeed[" << streams << "][3] = {\n"; /* resume table, reference main.cpp:200-207 */
            for (auto& s : simulate)
                for (auto& ch : s.channel) tout << setw(4) << '{' << ch.RS.IX << "," << ch.RS.IY << ',' << ch.RS.IZ << "},\n";
            tout << "}\n";
            tout.close();
            ++rounds;
            if (TestFrame > 1000 && ErrorFrame > 20) break;
            if (max_rounds && rounds >= max_rounds) break;
        }
        gettimeofday(&t_end, NULL);
        const double total_time = (t_end.tv_sec - t_start.tv_sec) + (t_end.tv_usec - t_start.tv_usec) / 1000000.0;
        unsigned long groups = 0;
        for (auto& s : simulate) { decode_s = decode_s > s.decode_seconds ? decode_s : s.decode_seconds; groups += s.decoded_groups; s.decoded_groups = 0; }
        const double gbps = decode_s > 0 ? (double)groups * 32 * NmoinsK / decode_s / 1e9 : 0;
        if (rank == 0) cout << setw(5) << snr << setw(20) << TestFrame << setw(15) << ErrorFrame << setw(20) << ErrorBits << setw(20) << FER << setw(20) << BER
             << setw(15) << LT3ErrBitFrame << setw(15) << total_time << setw(18) << gbps << endl;
        fout.open(rank == 0 ? "Result.txt" : "/dev/null", std::ios::app);
        fout << setw(5) << snr << '\t' << setw(20) << TestFrame << '\t' << setw(15) << ErrorFrame << '\t' << setw(20) << ErrorBits << '\t' << setw(20)
             << FER << '\t' << setw(20) << BER << '\t' << setw(15) << LT3ErrBitFrame << '\t' << setw(15) << total_time << '\t' << endl;
        fout.close();
        if (rank == 0) {
            /* reference main.cpp:183-186, :224-227: the demapper's own error rates.  The counters behind them are never
             * incremented in the reference either (the ModCalErr call is commented out, CSimulate.cpp:129), so the row is zeros */
            const unsigned long ModErrorBits = 0, ModErrorFrame = 0, ModErrorSymbol = 0;
            const double ModBER = (double)ModErrorBits / ((double)TestFrame * (NmoinsK - _ShortenBits));
            const double ModSER = (double)ModErrorSymbol / ((double)TestFrame * (NmoinsK - _ShortenBits) / p_simulation.mod_type);
            const double ModFER = (double)ModErrorFrame / TestFrame;
            ofstream demodout("demod.txt", std::ios::app);
            demodout << setw(5) << snr << '\t' << setw(20) << ModFER << '\t' << setw(20) << ModBER << '\t' << setw(20) << ModSER << '\t' << endl;
        }
    }
    return 0;
}
