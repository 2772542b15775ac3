/* commfile_selftest.cpp — the --comm-file rendezvous of lnsfaid_sim without a GPU (tests/test_host_commfile.py).
 *   commfile_selftest publish F RUNID [AGE_S]   write an id (bytes 0..127 xor AGE) as rank 0 would, AGE_S seconds in the past
 *   commfile_selftest fetch F RUNID TIMEOUT_MS  wait for this run's id as rank r > 0 would; prints its first byte; exit 3 on timeout
 *   commfile_selftest clear F */
#include <cstdlib>

#include "CommFile.h"

int main(int argc, char** argv)
{
    if (argc >= 3 && !strcmp(argv[1], "clear")) { commfile_clear(argv[2]); return 0; }
    if (argc >= 4 && !strcmp(argv[1], "publish")) {
        const int age = argc > 4 ? atoi(argv[4]) : 0;
        uint8_t id[COMMFILE_ID_BYTES];
        for (int i = 0; i < COMMFILE_ID_BYTES; ++i) id[i] = (uint8_t)(i ^ age);
        if (!commfile_publish(argv[2], argv[3], id)) return 2;
        if (age) { /* back-date the record: a leftover of an earlier run */
            FILE* f = fopen(argv[2], "r+b");
            CommFileRecord r;
            if (!f || fread(&r, sizeof(r), 1, f) != 1) return 2;
            r.written_at -= age;
            rewind(f);
            if (fwrite(&r, sizeof(r), 1, f) != 1) return 2;
            fclose(f);
        }
        return 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "fetch")) {
        uint8_t id[COMMFILE_ID_BYTES];
        const int rc = commfile_fetch(argv[2], argv[3], id, atoi(argv[4]), (int64_t)time(nullptr));
        if (rc) { fprintf(stderr, "no RCCL id for run '%s' in %s\n", argv[3], argv[2]); return 3; }
        printf("%d\n", (int)id[0]);
        return 0;
    }
    fprintf(stderr, "usage: %s publish|fetch|clear ...\n", argv[0]);
    return 2;
}
