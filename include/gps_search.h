// gps_search.h -- the reference's own search-stage API, re-implemented on the MI355X engine.
//
// Mirrors c/gps_offline.h:23-25,87-91 of JiaoXianjun/GNSS-GPS-SDR exactly (C++ linkage, no
// namespace, caller-defined globals), so that the reference's front end
// c/test_search_offline.cpp links against libgps_search.so unchanged (`make dropin-check`).
#ifndef GPS_SEARCH_H
#define GPS_SEARCH_H

extern double FC;      // carrier @ 2nd IF, Hz          (defined by the caller, c/test_search_offline.cpp:12)
extern double FS;      // sampling rate, Hz
extern double max_fo;  // Doppler search half-range, Hz

int SearchInit();                         // c/search_offline.cpp:74   returns 0, or a gpsacq error code
void SearchFree();                        // c/search_offline.cpp:114
void SearchTask(char *filename_1bit_bin); // c/search_offline.cpp:219  prints the per-run report to stdout
void SearchEnable(int sv);                // c/search_offline.cpp:213
int SearchCode(int sv, int g1);           // c/search_offline.cpp:205

// Not in the reference (its SearchTask is void and cannot fail after fopen): 0, or the gpsacq error code that made the last
// SearchTask() stop early (message on stderr).  A caller that does not know about it -- the reference's own main() -- loses nothing.
int SearchStatus();

#endif
