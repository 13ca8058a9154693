// program.hpp -- host-side translation of the reference's byte-code (polar_compiler.hh:51-58,
// executed by polar_decoder.hh:131-464) into the device instruction stream of sc_core.hpp.
//
// The reference interpreter keeps three running values while it walks the program -- the
// current level `lvl`, the partial-sum cursor `hrd` and the message cursor `msg`
// (polar_decoder.hh:133-134, updated at :140, :203, :204, :299, :331, :363, :364, :395, :396,
// :428, :459). They depend only on the program, so they are resolved here once per loaded
// program and the kernel reads them from constant memory instead of recomputing them.
#pragma once

#include <stdint.h>
#include <vector>

#include "construct.hpp"
#include "sc_core.hpp"

namespace polar {

struct HostProgram {
	int top = 0;        // level of the code, N = 1 << top
	uint32_t rows = 0;  // message symbols emitted per frame (K for a compiled frozen set)
	std::vector<DevOp> ops; // terminated by two OP_END words
};

enum { MAX_LEVEL = 24 };

// Returns 0 on success. A nonzero return marks a program on which the reference would trip
// one of its asserts (polar_decoder.hh:135, :170, :202, :463 ...) or run outside its arrays.
// Fuse runs "left|right|rate0_right at level l, then left|rate0_right at l-1, l-2, ..." (the
// descent into a child always starts with one of these, polar_compiler.hh:34-47) into OP_CHAIN
// words of depth <= max_depth (Group::CHAIN_DEPTH of the kernel variant). A stage whose output only feeds a rate0_right stage of the same
// chain is not stored: no later `right` reads that level (rate0_right replaces left + right).
inline void fuse_chains(std::vector<DevOp> &ops, int q, int max_depth)
{
	std::vector<DevOp> out;
	auto soft = [](const DevOp &d) { uint32_t o = d.op_level & 255; return o == OP_LEFT || o == OP_RIGHT || o == OP_RATE0_RIGHT; };
	for (size_t i = 0; i < ops.size();) {
		if (!soft(ops[i])) {
			out.push_back(ops[i++]);
			continue;
		}
		int lvl = (int)(ops[i].op_level >> 8), depth = 1;
		uint32_t first = ops[i].op_level & 255;
		uint32_t pad = (first == OP_LEFT ? 0u : first == OP_RIGHT ? 1u : 2u) << 3;
		// deeper only while every lane of the wave still owns a word of the last level: below
		// that, fusing trades parallel lanes for serial work in one lane
		while (depth < max_depth && i + depth < ops.size() && ((size_t)q << (lvl - depth - 1)) >= 64) {
			const DevOp &n = ops[i + depth];
			uint32_t o = n.op_level & 255;
			if ((o != OP_LEFT && o != OP_RATE0_RIGHT) || (int)(n.op_level >> 8) != lvl - depth)
				break;
			++depth;
			if (o == OP_RATE0_RIGHT)
				pad |= 1u << (4 + depth) | 1u << (12 + depth - 1);
		}
		DevOp c = {OP_CHAIN | (uint32_t)lvl << 8, ops[i].hrd, 0, pad | (uint32_t)depth};
		out.push_back(c);
		i += depth;
	}
	ops.swap(out);
}

// Replace every composite node of level 4, else 3 (left|rate0_right ... comb|rate0_comb|
// rate1_comb and all nodes in between) by one OP_SUBTREE word carrying the frozen mask of its
// positions. The device derives the node decomposition from the mask by the compiler's rules, so a
// node is fused only if compiling its mask gives back exactly its bytes (always the case for
// programs of PolarCompiler; a hand-made program that decomposes a node differently keeps its ops).
// ops[i] corresponds to prog[1 + i].
inline void fuse_subtrees(std::vector<DevOp> &ops, const uint8_t *prog)
{
	std::vector<DevOp> out;
	auto level_of = [](const DevOp &d) { return (int)(d.op_level >> 8); };
	// the composite node of level L that starts at ops[i]: its last op, or 0 if it cannot be fused
	auto try_fuse = [&](size_t i, int L, DevOp &fused) -> size_t {
		size_t e = i + 1; // the comb that completes this node: the first one at level L
		for (; e < ops.size(); ++e) {
			uint32_t oe = ops[e].op_level & 255;
			if ((oe == OP_COMB || oe == OP_RATE0_COMB || oe == OP_RATE1_COMB) && level_of(ops[e]) == L)
				break;
		}
		if (e >= ops.size())
			return 0;
		uint8_t frozen[16] = {};
		const uint32_t base = ops[e].hrd;
		uint32_t first_row = ~0u;
		for (size_t k = i; k <= e; ++k) {
			const uint32_t o = ops[k].op_level & 255, at = ops[k].hrd - base, n = 1u << level_of(ops[k]);
			bool emits = false;
			switch (o) {
			case OP_RATE0: for (uint32_t j = 0; j < n; ++j) frozen[at + j] = 1; break;
			case OP_REP: for (uint32_t j = 0; j + 1 < n; ++j) frozen[at + j] = 1; emits = true; break;
			case OP_SPC: frozen[at] = 1; emits = true; break;
			case OP_RATE1: emits = true; break;
			case OP_RATE1_COMB: emits = true; break;
			case OP_RATE0_COMB: for (uint32_t j = 0; j < n / 2; ++j) frozen[at + j] = 1; break;
			default: break;
			}
			if (emits && first_row == ~0u)
				first_row = ops[k].row;
		}
		uint8_t bytes[64]; // would the compiler decompose this mask into these very bytes?
		ProgramBuilder b;
		if (b.run(bytes, frozen, L) != (int)(e - i + 1) + 2)
			return 0;
		for (size_t k = i; k <= e; ++k)
			if (bytes[1 + (k - i)] != prog[1 + k])
				return 0;
		uint32_t mask = 0;
		for (int j = 0; j < (1 << L); ++j)
			mask |= (uint32_t)frozen[j] << j;
		fused = DevOp{OP_SUBTREE | (uint32_t)L << 8, base, first_row == ~0u ? 0u : first_row, mask};
		return e;
	};
	for (size_t i = 0; i < ops.size();) {
		const uint32_t o = ops[i].op_level & 255;
		const int L = level_of(ops[i]);
		DevOp fused;
		size_t e = 0;
		if ((o == OP_LEFT || o == OP_RATE0_RIGHT) && (L == 4 || L == 3))
			e = try_fuse(i, L, fused);
		if (e) {
			out.push_back(fused);
			i = e + 1;
		} else {
			out.push_back(ops[i++]);
		}
	}
	ops.swap(out);
}

// fuse_q: frame quads per wavefront of the kernel that will run the program (0 = no fusion)
inline int translate_program(const uint8_t *prog, int length, HostProgram &out, int fuse_q = 0, int max_depth = CHAIN_MAX)
{
	if (length < 2)
		return 1;
	int top = prog[0];
	if (top < 1 || top > MAX_LEVEL)
		return 2;
	int lvl = top;
	uint32_t n = 1u << top, hrd = 0, row = 0;
	out.top = top;
	out.ops.clear();
	int pc = 1;
	for (;; ++pc) {
		if (pc >= length)
			return 3; // no terminator
		uint32_t op = prog[pc];
		if (op == OP_END)
			break;
		DevOp d = {0, 0, 0, 0};
		switch (op) {
		case OP_LEFT:
		case OP_RATE0_RIGHT:
			if (lvl < 2)
				return 4;
			d.op_level = op | (uint32_t)lvl << 8;
			--lvl;
			if (op == OP_RATE0_RIGHT)
				hrd += 1u << lvl;
			break;
		case OP_RIGHT:
			if (lvl + 1 > top)
				return 5;
			d.op_level = op | (uint32_t)(lvl + 1) << 8;
			d.hrd = hrd; // the decided left child
			hrd += 1u << lvl;
			break;
		case OP_COMB:
		case OP_RATE0_COMB:
			if (lvl + 1 > top || hrd < (1u << lvl))
				return 6;
			hrd -= 1u << lvl;
			++lvl;
			d.op_level = op | (uint32_t)lvl << 8;
			d.hrd = hrd;
			break;
		case OP_RATE1_COMB:
			if (lvl + 1 > top)
				return 7;
			++lvl;
			d.op_level = op | (uint32_t)lvl << 8;
			d.hrd = hrd;
			d.row = row;
			row += 1u << (lvl - 1);
			break;
		case OP_RATE0:
		case OP_RATE1:
		case OP_REP:
		case OP_SPC:
			d.op_level = op | (uint32_t)lvl << 8;
			d.hrd = hrd;
			d.row = row;
			row += op == OP_RATE1 ? 1u << lvl : op == OP_REP ? 1u : op == OP_SPC ? (1u << lvl) - 1 : 0u;
			break;
		default:
			return 8;
		}
		int node = (int)(d.op_level >> 8);
		// the node must sit on its natural boundary inside the code
		if (d.hrd & ((1u << node) - 1) || d.hrd + (1u << node) > n || row > n)
			if (op != OP_LEFT && op != OP_RATE0_RIGHT)
				return 9;
		out.ops.push_back(d);
	}
	if (lvl != top)
		return 10;
	if (fuse_q) {
		// Narrow groups are bound by the per-node latency of the small nodes, not by lanes: there one
		// lane per frame quad decoding a whole level-3 node in registers beats five ops over LDS
		// (+5...11 % at 8 Ki ... 32 Ki frames); with 32 quads per wavefront the LDS passes keep all 64
		// lanes busy and the in-register form leaves half of them idle (-7 % at N = 2^10, measured).
		if (fuse_q <= SUBTREE_MAX_Q)
			fuse_subtrees(out.ops, prog);
		fuse_chains(out.ops, fuse_q, max_depth < CHAIN_MAX ? max_depth : (int)CHAIN_MAX);
	}
	DevOp e = {OP_END, 0, 0, 0};
	out.ops.push_back(e);
	out.ops.push_back(e); // the interpreter fetches one word ahead
	out.rows = row;
	return 0;
}

} // namespace polar
