// tools/dev/irw_prims.hip -- the lane primitives of k_intra_rows (enc_k_intra.hip) against their scalar meaning, on the GPU:
// the boustrophedon transforms, the quantiser on the signed product, colpack / rowpack / row_bcast:15, and the predictor
// table (two v_perm_b32 + two v_sad_u8 + shift) against pred4_px for every mode, availability case and lane.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I cedarx_h264_encoder_amd/csrc tools/dev/irw_prims.hip -o tools/dev/irw_prims
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../cedarx_h264_encoder_amd/csrc/enc_k_intra.hip"

using namespace cedar;

namespace cedar {
void build_pred4_table(uint8_t tab[9 * 16])
{
	for (int y = 0; y < 4; y++)
		for (int x = 0; x < 4; x++) {
			const int i = 4 * y + x;
			auto set = [&](int mode, int kind, int k) { tab[mode * 16 + i] = (uint8_t)((kind << 4) | k); };
			set(0, 0, 6 + x);
			set(1, 0, 4 - y);
			set(2, 3, 0);
			set(3, 2, 7 + x + y);
			set(4, 2, 5 + x - y);
			{
				int z = 2 * x - y;
				if (z >= 0 && !(z & 1)) set(5, 1, 5 + x - (y >> 1));
				else if (z >= 0) set(5, 2, 5 + x - (y >> 1));
				else if (z == -1) set(5, 2, 5);
				else set(5, 2, 6 - y);
			}
			{
				int z = 2 * y - x;
				if (z >= 0 && !(z & 1)) set(6, 1, 4 - y + (x >> 1));
				else if (z >= 0) set(6, 2, 5 - y + (x >> 1));
				else if (z == -1) set(6, 2, 5);
				else set(6, 2, 4 + x);
			}
			if (!(y & 1)) set(7, 1, 6 + x + (y >> 1));
			else set(7, 2, 7 + x + (y >> 1));
			{
				int z = x + 2 * y, k = y + (x >> 1);
				if (z > 5) set(8, 0, 1);
				else if (z == 5) set(8, 2, 1);
				else if (!(z & 1)) set(8, 1, 3 - k);
				else set(8, 2, 3 - k);
			}
		}
}
}

// out[0..63]: fdct; [64..127]: idct of in2; [128..191] quant; [192..255] colpack; [256..319] rowpack; [320..383] bcast
__global__ void k_prims(const int *in, const int *in2, int *out, int mf, int f, int qbits)
{
	const int lane = threadIdx.x;
	const IrwLane L = irw_lane(lane & 15);
	const int g = lane >> 4;
	const int v = in[g * 16 + 4 * L.qi + L.qj]; // sample (i, j) of block g
	out[lane] = irw_fdct(v, L.ci, L.cj);
	const int kv = L.qi == 1 ? 2 : L.qi == 2 ? 1 : L.qi, kh = L.qj == 1 ? 2 : L.qj == 2 ? 1 : L.qj;
	const int d = in2[g * 16 + 4 * kv + kh] + (L.dc_lane ? 32 : 0);
	out[64 + lane] = irw_idct(d, L.ci, L.cj);
	out[128 + lane] = irw_quant(in2[lane] - 20000, mf, f, (1 << qbits) - 1 - 2 * f, qbits);
	const int rec = v & 255;
	{
		int q = __builtin_amdgcn_update_dpp(0, rec, 0xFF, 0xf, 0x5, false);
		q = __builtin_amdgcn_update_dpp(q, rec, 0x00, 0xf, 0xA, false);
		uint32_t z = (uint32_t)q << (8 * L.qi);
		z |= (uint32_t)IRW_DPP(z, 0x124);
		z |= (uint32_t)IRW_DPP(z, 0x128);
		out[192 + lane] = (int)z;
	}
	uint32_t y = (uint32_t)rec << (8 * L.qj);
	y |= (uint32_t)IRW_DPP(y, 0x111);
	y |= (uint32_t)IRW_DPP(y, 0x112);
	out[256 + lane] = (int)y;
	out[320 + lane] = __builtin_amdgcn_update_dpp(0x12345678, (int)y, 0x142, 0xE, 0xF, false);
}

__global__ void k_pred(const uint8_t *tab, const uint8_t *edges, int *out)
{
	// block b of the launch: predictor row r = b % 22, edge set b / 22; lane = sample
	const int lane = threadIdx.x & 15, row = blockIdx.x % IRW_TAB_ROWS, es = blockIdx.x / IRW_TAB_ROWS;
	const IrwLane L = irw_lane(lane);
	const IrwTabEnt t = irw_tab_entry(tab, row, L.qi, L.qj);
	const uint8_t *E = edges + es * 16; // E[1..13]
	const uint32_t Lc = E[4] | E[3] << 8 | E[2] << 16 | (uint32_t)E[1] << 24;
	const uint32_t P2 = (uint32_t)E[5] << 24;
	const uint32_t P1 = E[6] | E[7] << 8 | E[8] << 16 | (uint32_t)E[9] << 24;
	const uint32_t A = E[10] | E[11] << 8 | E[12] << 16 | (uint32_t)E[13] << 24;
	const uint32_t pa = __builtin_amdgcn_perm(P2, Lc, t.sel_a), pb = __builtin_amdgcn_perm(A, P1, t.sel_b);
	out[blockIdx.x * 16 + 4 * L.qi + L.qj] = (int)(__builtin_amdgcn_sad_u8(pb, 0u, __builtin_amdgcn_sad_u8(pa, 0u, t.rnd)) >> t.sh);
}

static void fdct_ref(const int r[16], int w[16])
{
	int t[16];
	for (int i = 0; i < 4; i++) {
		int a = r[4 * i] + r[4 * i + 3], b = r[4 * i + 1] + r[4 * i + 2], c = r[4 * i + 1] - r[4 * i + 2], d = r[4 * i] - r[4 * i + 3];
		t[4 * i] = a + b; t[4 * i + 1] = 2 * d + c; t[4 * i + 2] = a - b; t[4 * i + 3] = d - 2 * c;
	}
	for (int i = 0; i < 4; i++) {
		int a = t[i] + t[12 + i], b = t[4 + i] + t[8 + i], c = t[4 + i] - t[8 + i], d = t[i] - t[12 + i];
		w[i] = a + b; w[4 + i] = 2 * d + c; w[8 + i] = a - b; w[12 + i] = d - 2 * c;
	}
}
static void idct_ref(const int d[16], int out[16])
{
	int t[16];
	for (int i = 0; i < 4; i++) {
		int e0 = d[4 * i] + d[4 * i + 2], e1 = d[4 * i] - d[4 * i + 2], e2 = (d[4 * i + 1] >> 1) - d[4 * i + 3], e3 = d[4 * i + 1] + (d[4 * i + 3] >> 1);
		t[4 * i] = e0 + e3; t[4 * i + 1] = e1 + e2; t[4 * i + 2] = e1 - e2; t[4 * i + 3] = e0 - e3;
	}
	for (int i = 0; i < 4; i++) {
		int e0 = t[i] + t[8 + i], e1 = t[i] - t[8 + i], e2 = (t[4 + i] >> 1) - t[12 + i], e3 = t[4 + i] + (t[12 + i] >> 1);
		out[i] = (e0 + e3 + 32) >> 6; out[4 + i] = (e1 + e2 + 32) >> 6; out[8 + i] = (e1 - e2 + 32) >> 6; out[12 + i] = (e0 - e3 + 32) >> 6;
	}
}
static int pred_ref(int ent, const uint8_t *E15, int dc)
{
	int kind = ent >> 4, k = ent & 15;
	int a = E15[k > 0 ? k - 1 : 0], b = E15[k], c = E15[k + 1 < 15 ? k + 1 : 14];
	return kind == 0 ? b : kind == 1 ? (b + c + 1) >> 1 : kind == 2 ? (a + 2 * b + c + 2) >> 2 : dc;
}

int main()
{
	int bad = 0;
	srand(7);
	std::vector<int> in(64), in2(64), out(384);
	for (int rep = 0; rep < 20; rep++) {
		for (int i = 0; i < 64; i++) {
			in[i] = rand() % 511 - 255;
			in2[i] = rand() % 40001 - 20000 + 20000; // quant input is in2 - 20000
		}
		std::vector<int> d2(64);
		for (int i = 0; i < 64; i++)
			d2[i] = rand() % 8001 - 4000;
		int *din, *din2, *dout;
		hipMalloc(&din, 256); hipMalloc(&din2, 256); hipMalloc(&dout, 384 * 4);
		const int qbits = 15 + rep % 8, mf = 13107 - 700 * (rep % 5), f = (1 << qbits) / 3;
		// idct and quant share in2: run twice
		hipMemcpy(din, in.data(), 256, hipMemcpyHostToDevice);
		hipMemcpy(din2, d2.data(), 256, hipMemcpyHostToDevice);
		k_prims<<<1, 64>>>(din, din2, dout, mf, f, qbits);
		hipMemcpy(out.data(), dout, 384 * 4, hipMemcpyDeviceToHost);
		for (int g = 0; g < 4; g++) {
			int w[16], r[16];
			fdct_ref(&in[g * 16], w);
			idct_ref(&d2[g * 16], r);
			for (int l = 0; l < 16; l++) {
				const int qi = l >> 2, qp = l & 3, qj = (qi & 1) ? 3 - qp : qp;
				const int kv = qi == 1 ? 2 : qi == 2 ? 1 : qi, kh = qj == 1 ? 2 : qj == 2 ? 1 : qj;
				if (out[g * 16 + l] != w[4 * kv + kh]) { bad++; if (bad < 10) printf("fdct g%d l%d: %d want %d\n", g, l, out[g * 16 + l], w[4 * kv + kh]); }
				if (out[64 + g * 16 + l] != r[4 * qi + qj]) { bad++; if (bad < 10) printf("idct g%d l%d: %d want %d\n", g, l, out[64 + g * 16 + l], r[4 * qi + qj]); }
				// colpack: byte i = sample (i, 3) of the group's block; rowpack in lane 15: byte j = sample (3, j)
				uint32_t col = 0, rowp = 0;
				for (int k = 0; k < 4; k++) {
					col |= (uint32_t)(in[g * 16 + 4 * k + 3] & 255) << (8 * k);
					rowp |= (uint32_t)(in[g * 16 + 12 + k] & 255) << (8 * k);
				}
				if ((uint32_t)out[192 + g * 16 + l] != col) { bad++; if (bad < 10) printf("colpack g%d l%d: %08x want %08x\n", g, l, out[192 + g * 16 + l], col); }
				if (l == 15 && (uint32_t)out[256 + g * 16 + l] != rowp) { bad++; if (bad < 10) printf("rowpack g%d: %08x want %08x\n", g, out[256 + g * 16 + l], rowp); }
				uint32_t want_b = 0x12345678u;
				if (g > 0) {
					want_b = 0;
					for (int k = 0; k < 4; k++)
						want_b |= (uint32_t)(in[(g - 1) * 16 + 12 + k] & 255) << (8 * k);
				}
				if ((uint32_t)out[320 + g * 16 + l] != want_b) { bad++; if (bad < 10) printf("bcast g%d l%d: %08x want %08x\n", g, l, out[320 + g * 16 + l], want_b); }
			}
		}
		for (int i = 0; i < 64; i++) {
			const int w = d2[i] - 20000 + 0; // k_prims quantises in2[lane] - 20000
			const int a = abs(w);
			long long lv = ((long long)a * mf + f) >> qbits;
			if (lv > 2047) lv = 2047;
			const int want = w < 0 ? -(int)lv : (int)lv;
			if (out[128 + i] != want) { bad++; if (bad < 10) printf("quant %d (w %d mf %d q %d): %d want %d\n", i, w, mf, qbits, out[128 + i], want); }
		}
		hipFree(din); hipFree(din2); hipFree(dout);
	}
	// predictors
	uint8_t tab[144];
	build_pred4_table(tab);
	const int NE = 50;
	std::vector<uint8_t> edges(NE * 16);
	for (auto &e : edges)
		e = (uint8_t)(rand() & 255);
	uint8_t *dtab, *dedges;
	int *dout;
	hipMalloc(&dtab, 144); hipMalloc(&dedges, NE * 16); hipMalloc(&dout, NE * IRW_TAB_ROWS * 16 * 4);
	hipMemcpy(dtab, tab, 144, hipMemcpyHostToDevice);
	hipMemcpy(dedges, edges.data(), NE * 16, hipMemcpyHostToDevice);
	k_pred<<<NE * IRW_TAB_ROWS, 16>>>(dtab, dedges, dout);
	std::vector<int> po(NE * IRW_TAB_ROWS * 16);
	hipMemcpy(po.data(), dout, po.size() * 4, hipMemcpyDeviceToHost);
	for (int es = 0; es < NE; es++)
		for (int row = 0; row < IRW_TAB_ROWS; row++) {
			const uint8_t *E = &edges[es * 16];
			uint8_t E15[15];
			for (int k = 1; k <= 13; k++)
				E15[k] = E[k];
			int mode, dc = 128;
			if (row >= 18) {
				mode = 2;
				const bool top = (row - 18) & 1, left = (row - 18) & 2;
				const int st = E[6] + E[7] + E[8] + E[9], sl = E[1] + E[2] + E[3] + E[4];
				dc = top && left ? (st + sl + 4) >> 3 : left ? (sl + 2) >> 2 : top ? (st + 2) >> 2 : 128;
			} else {
				mode = row % 9;
				if (mode == 2)
					continue;
				if (row >= 9)
					for (int k = 10; k <= 13; k++)
						E15[k] = E15[9];
			}
			E15[0] = E15[1];
			E15[14] = E15[13];
			for (int i = 0; i < 16; i++) {
				const int want = pred_ref(tab[mode * 16 + i], E15, dc);
				const int got = po[(es * IRW_TAB_ROWS + row) * 16 + i];
				if (got != want) { bad++; if (bad < 30) printf("pred row %d sample %d: %d want %d\n", row, i, got, want); }
			}
		}
	printf(bad ? "irw_prims: %d MISMATCHES\n" : "irw_prims: all primitives equal their scalar meaning (%d)\n", bad);
	return bad ? 1 : 0;
}
