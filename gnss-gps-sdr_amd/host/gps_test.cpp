// gps_test -- command-line front end of the MI355X acquisition engine with the command-line
// surface and stdout of the reference's front end (c/test_search_offline.cpp:15-49):
//
//     gps_test [capture carrier_freq sampling_rate max_freq_offset]
//
// Behaviour kept on purpose: the usage text is printed on every start; exactly 0 or 4 arguments
// are accepted, anything else prints one line and exits 0; the fourth argument is accepted but not
// used -- the search range stays +-5 kHz, as in the reference (:22,31-34) -- unless
// GPSACQ_HONOR_MAX_FO=1 is set.  A failing SearchInit() is reported through the exit status.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/gps_search.h"

// the three globals the search stage reads (declared extern in gps_search.h / c/gps_offline.h:23-25)
double FC = 4.092e6, FS = 5.456e6, max_fo = 5000.0;

static const char kUsage[] =
    "GPS CA code offline search. Extract from http://www.aholme.co.uk/GPS/Main.htm\n"
    "Jiao Xianjun (putaoshu@gmail.com). 2014-05.\n"
    "usage:\n"
    "gps_test   filename_of_1bit_IF_cap   carrier_freq   sampling_rate   max_freq_offset\n"
    "or\n"
    "gps_test (Make sure gps.samples.1bit.I.fs5456.if4092.bin can be found. Download http://www.jks.com/gps/gps.html)\n";

static bool env_flag(const char *name) {
    const char *v = std::getenv(name);
    return v != nullptr && std::atoi(v) != 0;
}

int main(int argc, char **argv) {
    std::string capture = "gps.samples.1bit.I.fs5456.if4092.bin";  // the Nottingham capture, :19
    std::fputs(kUsage, stdout);

    const int n_args = argc - 1;
    if (n_args == 4) {
        capture = argv[1];
        FC = std::atof(argv[2]);
        FS = std::atof(argv[3]);
        if (env_flag("GPSACQ_HONOR_MAX_FO")) max_fo = std::atof(argv[4]);
    } else if (n_args != 0) {
        std::puts("Please run with 3 arguments or without argument!");
        return 0;
    }

    if (const int rc = SearchInit()) {
        std::printf("SearchInit() returned %d\n", rc);
        return rc;
    }
    std::fflush(stdout);
    SearchTask(&capture[0]);
    const int status = SearchStatus();  // a device failure in mid-file: nonzero exit (the reference cannot fail there)
    SearchFree();
    return status;
}
