// bfio.h -- btllib Bloom-filter file header IO (host side).
//
// Format (btllib BloomFilter::save, restated from its published behaviour --
// btllib itself is not in the reference tree, so this is "parity unpinned",
// see DESIGN.md): a TOML-ish text header
//     [BTLKmerBloomFilter_vN]          (or [BTLKmerCountingBloomFilter_vN])
//     bytes = <array bytes>
//     hash_num = <h>
//     hash_fn = "ntHash_v2"
//     k = <k>
//     [HeaderEnd]
// followed by the raw array.  Keys may come in any order (cpptoml tables are
// unordered); unknown keys are ignored; any _vN suffix is accepted.
#pragma once
#include <cstdint>
#include <cstdio>

namespace nte_host {

struct BfHeader
{
	uint64_t bytes = 0;
	uint32_t hash_num = 0;
	uint32_t k = 0;
	bool counting = false;
	char hash_fn[32] = { 0 }; // as written in the header ("ntHash_v2" for every filter ntEdit can use)
};

// opens the file and parses the header; on success the returned FILE* is
// positioned at the first byte of the array.  nullptr on any error; why (when not nullptr) then says
// what was wrong with the file.
FILE* bf_open(const char* path, BfHeader* h, const char** why = nullptr);
// writes header + array; 0 on success
int bf_save(const char* path, const BfHeader& h, const uint8_t* data);

} // namespace nte_host
