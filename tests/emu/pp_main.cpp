// pp_main.cpp -- test harness for csrc/bam_header.hpp: reads a BAM text header on standard input and prints the PP: value and the ID:
// the new @PG line gets (find_pp_tag), then -- one per line -- the @PG IDs in the slot order of the string set.
#include <stdio.h>
#include <iostream>
#include <iterator>
#include "../../network-aware-bwa_amd/csrc/bam_header.hpp"

int main(int argc, char **argv)
{
	std::string h((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
	if (argc > 1) {                          /* "--slots": the words on standard input (one per line) in slot order */
		WordSlots w; size_t p = 0;
		while (p < h.size()) { size_t e = h.find('\n', p); if (e == std::string::npos) e = h.size(); w.add(h.substr(p, e - p)); p = e + 1; }
		for (size_t i = 0; i < w.key.size(); ++i) if (w.full[i]) printf("%s\n", w.key[i].c_str());
		return 0;
	}
	std::string pp, id; bool has;
	find_pp_tag(h, pp, id, has);
	printf("%s\n%s\n", has ? pp.c_str() : "-", id.c_str());
	return 0;
}
