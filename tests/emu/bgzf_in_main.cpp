// bgzf_in_main.cpp -- test harness for csrc/bgzf_in.hpp: the plain bytes of the file named on the command line go to standard
// output, read in the uneven steps a BAM reader takes (4 bytes, then a record); a damaged file ends with status 3 and a message.
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void die(const char *what, const char *why) { fprintf(stderr, "%s: %s\n", what, why); fflush(stderr); _exit(3); }
#include "../../network-aware-bwa_amd/csrc/bgzf_in.hpp"

int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f) return 2;
	BamIn in(f, argv[1]);
	std::vector<uint8_t> buf;
	size_t step = 1;
	for (;;) {
		const size_t want = 4 + step % 70001;            /* small and large reads, across every block boundary sooner or later */
		const size_t have = in.need(want);
		if (have == 0) break;
		const size_t n = have < want ? have : want;
		buf.resize(n);
		if (!in.read(buf.data(), n)) return 4;
		if (fwrite(buf.data(), 1, n, stdout) != n) return 5;
		step = step * 31 + 7;
	}
	fprintf(stderr, "%s\n", in.bgzf ? "bgzf" : in.raw ? "raw" : "stream");
	return 0;
}
