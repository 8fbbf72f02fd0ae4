// metdet_host_only.cpp -- the host-only entry points of libmetdet_hip.so as a library of their own, for the sanitizer build
// (`make san` -> ../libmetdet_host_san.so).  Exports mdh_last_error, mdh_abi_version, mdh_lineset_nms and mdh_collector_*;
// everything that needs a GPU is absent (the Python binding reports a host-only build when such an entry is called).
#include "metdet_host.hpp"
