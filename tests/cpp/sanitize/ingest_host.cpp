// ingest_host.cpp — the graph-JSON ingest's C ABI (sgtd_graphs_*) compiled WITHOUT HIP: the same two headers
// sgtd_accel.hip includes, as a translation unit of their own for the sanitizer builds (ASan + UBSan) of
// fuzz_files.cpp and of tests/cpp/test_ingest_nlohmann.cpp.
#include "../../../sgtd_amd/csrc/graph_ingest_abi.h"
