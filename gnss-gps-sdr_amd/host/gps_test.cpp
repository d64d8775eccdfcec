// gps_test -- command-line front end with the surface of the reference's
// c/test_search_offline.cpp:15-49: `gps_test [file carrier_freq sampling_rate max_freq_offset]`.
// Like the reference, the fourth argument is accepted but not read (max_fo stays 5000 Hz,
// :22,31-34) unless GPSACQ_HONOR_MAX_FO=1 is set in the environment.
#include <cstdio>
#include <cstdlib>
#include <cstring>

double FC, FS, max_fo;
#include "../../include/gps_search.h"

int main(int argc, char *argv[]) {
    char filename[4096];
    snprintf(filename, sizeof filename, "%s", "gps.samples.1bit.I.fs5456.if4092.bin");
    FC = 4.092e6;
    FS = 5.456e6;
    max_fo = 5000.0;

    printf("GPS CA code offline search. Extract from http://www.aholme.co.uk/GPS/Main.htm\n");
    printf("Jiao Xianjun (putaoshu@gmail.com). 2014-05.\n");
    printf("usage:\n");
    printf("gps_test   filename_of_1bit_IF_cap   carrier_freq   sampling_rate   max_freq_offset\n");
    printf("or\n");
    printf("gps_test (Make sure gps.samples.1bit.I.fs5456.if4092.bin can be found. Download http://www.jks.com/gps/gps.html)\n");

    if (argc == 5) {
        snprintf(filename, sizeof filename, "%s", argv[1]);
        FC = atof(argv[2]);
        FS = atof(argv[3]);
        const char *honor = getenv("GPSACQ_HONOR_MAX_FO");
        if (honor && atoi(honor) != 0) max_fo = atof(argv[4]);
    } else if (argc != 1) {
        printf("Please run with 3 arguments or without argument!\n");
        return 0;
    }

    int ret = SearchInit();
    if (ret) {
        printf("SearchInit() returned %d\n", ret);
        return ret;
    }
    fflush(stdout);
    SearchTask(filename);
    SearchFree();
    return 0;
}
