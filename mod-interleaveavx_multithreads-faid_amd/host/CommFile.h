/*
 * CommFile.h — hand the 128-byte RCCL id of a multi-process run from rank 0 to the other ranks through a file
 * (`lnsfaid_sim --ranks N --rank r --comm-file F`).  The reference has no process boundary (one pthread per worker,
 * counters summed after pthread_join, reference main.cpp:166-182); this is the rendezvous of its one-process-per-GPU
 * counterpart, in front of lnsfaid_comm_init.
 *
 * A file left by an earlier run must never be taken for this run's id: ncclCommInitRank with a dead id blocks for ever
 * with the GPU held.  So the file carries a header {magic, hash of --run-id, rank 0's start time}, and
 *   rank 0    removes F and F.tmp before it creates the id, publishes by rename (readers never see a partial file), and
 *             removes F again once lnsfaid_comm_init has returned (it returns when every rank has joined);
 *   rank r>0  accepts a file only if magic and run id match and it was written no more than `max_age_s` before the
 *             reader itself started (a leftover of a crashed run of the same --run-id: pick a fresh id per run); anything else
 *             is polled over; the wait is bounded and ends in a non-zero exit.
 */
#ifndef LNSFAID_COMMFILE_H
#define LNSFAID_COMMFILE_H

#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <string>

#define COMMFILE_ID_BYTES 128
#define COMMFILE_MAGIC 0x4c4e5346414944ull /* "LNSFAID" */

struct CommFileRecord {
    uint64_t magic, run_hash;
    int64_t written_at; /* rank 0's clock, seconds since the epoch */
    uint8_t id[COMMFILE_ID_BYTES];
};

static inline uint64_t commfile_hash(const char* run_id)
{
    uint64_t h = 1469598103934665603ull; /* FNV-1a */
    for (const unsigned char* p = (const unsigned char*)(run_id ? run_id : ""); *p; ++p) h = (h ^ *p) * 1099511628211ull;
    return h;
}

/* rank 0, before it creates the id */
static inline void commfile_clear(const char* path)
{
    unlink(path);
    unlink((std::string(path) + ".tmp").c_str());
}

/* rank 0: write the record to F.tmp, rename over F */
static inline bool commfile_publish(const char* path, const char* run_id, const uint8_t id[COMMFILE_ID_BYTES])
{
    CommFileRecord r;
    r.magic = COMMFILE_MAGIC; r.run_hash = commfile_hash(run_id); r.written_at = (int64_t)time(nullptr);
    memcpy(r.id, id, COMMFILE_ID_BYTES);
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(&r, sizeof(r), 1, f) == 1;
    if (fclose(f) != 0 || !ok) { unlink(tmp.c_str()); return false; }
    return rename(tmp.c_str(), path) == 0;
}

/* rank r > 0: 0 = got this run's id; 1 = timed out (nothing acceptable appeared) */
static inline int commfile_fetch(const char* path, const char* run_id, uint8_t id[COMMFILE_ID_BYTES], int timeout_ms,
                                 int64_t reader_started_at, int max_age_s = 300)
{
    const uint64_t want = commfile_hash(run_id);
    for (int waited = 0;; waited += 50) {
        FILE* f = fopen(path, "rb");
        if (f) {
            CommFileRecord r;
            const bool whole = fread(&r, sizeof(r), 1, f) == 1;
            fclose(f);
            if (whole && r.magic == COMMFILE_MAGIC && r.run_hash == want && r.written_at + max_age_s >= reader_started_at) {
                memcpy(id, r.id, COMMFILE_ID_BYTES);
                return 0;
            }
        }
        if (waited >= timeout_ms) return 1;
        usleep(50 * 1000);
    }
}

#endif
