// kwage_amd/csrc/kmer_device.hpp -- device helpers shared by the search kernels (kernels.hpp) and the
// Bloom construction kernels (counter.hip): 2-bit base codes, reverse complement, and MurmurHash3_x86_32
// over the ASCII k-mer with a seed-independent key schedule.
#ifndef KWAGE_AMD_KMER_DEVICE_HPP
#define KWAGE_AMD_KMER_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kwage {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
	h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
	return h;
}

// MurmurHash3_x86_32 over the k ASCII bases of `word` (most significant base first),
// reference hash.cpp:176-234.  The per-block key schedule does not depend on the seed, so it is
// computed once and reused for every seed.
struct MurmurKeys {
	uint32_t k1[8];      // mixed 4-byte blocks (k <= 32 -> at most 8)
	uint32_t tail;       // mixed tail (0 if k % 4 == 0)
};

__device__ __forceinline__ uint32_t base_ascii(uint64_t word, uint32_t k, uint32_t i)
{
	// "ACGT"[code] packed as bytes 0x41,0x43,0x47,0x54 (reference word.h:31-34)
	const uint32_t code = (uint32_t)(word >> (2*(k - 1 - i))) & 3u;
	return (0x54474341u >> (8*code)) & 0xFFu;
}

__device__ __forceinline__ void murmur_keys(uint64_t word, uint32_t k, MurmurKeys &mk)
{
	const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
	const uint32_t nblocks = k >> 2;
#pragma unroll
	for(uint32_t b = 0; b < 8; ++b){
		uint32_t v = 0;
		if(b < nblocks){
			v = base_ascii(word, k, 4*b) | (base_ascii(word, k, 4*b + 1) << 8) |
			    (base_ascii(word, k, 4*b + 2) << 16) | (base_ascii(word, k, 4*b + 3) << 24);
			v *= c1; v = rotl32(v, 15); v *= c2;
		}
		mk.k1[b] = v;
	}
	uint32_t t = 0;
	const uint32_t off = nblocks*4;
	switch(k & 3u){
		case 3: t ^= base_ascii(word, k, off + 2) << 16; [[fallthrough]];
		case 2: t ^= base_ascii(word, k, off + 1) << 8;  [[fallthrough]];
		case 1: t ^= base_ascii(word, k, off);
			t *= c1; t = rotl32(t, 15); t *= c2;
	}
	mk.tail = t;
}

__device__ __forceinline__ uint32_t murmur_finish(const MurmurKeys &mk, uint32_t k, uint32_t seed)
{
	const uint32_t nblocks = k >> 2;
	uint32_t h1 = seed;
#pragma unroll
	for(uint32_t b = 0; b < 8; ++b){
		if(b < nblocks){
			h1 ^= mk.k1[b]; h1 = rotl32(h1, 13); h1 = h1*5u + 0xe6546b64u;
		}
	}
	h1 ^= mk.tail;      // zero when k % 4 == 0, exactly like skipping the tail switch
	h1 ^= k;
	return fmix32(h1);
}

// Reverse complement of the low 2k bits (A=0,C=1,G=2,T=3 so complement == bitwise NOT);
// equals the reference's rolling __comp_w & mask (word.h:87-99,164).
__device__ __forceinline__ uint64_t revcomp2(uint64_t w, uint32_t k)
{
	uint64_t x = ~w;
	x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
	x = __builtin_bswap64(x);
	return x >> (64 - 2*k);
}

__device__ __forceinline__ uint32_t base_code(char ch)
{
	// word.h:84-104: ACGT in either case are bases, anything else resets the run.
	switch(ch){
		case 'A': case 'a': return 0;
		case 'C': case 'c': return 1;
		case 'G': case 'g': return 2;
		case 'T': case 't': return 3;
		default: return 4;
	}
}

}  // namespace kwage

#endif
