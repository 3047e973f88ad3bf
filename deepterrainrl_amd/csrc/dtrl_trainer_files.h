// dtrl_trainer_files.h -- host side: a trainer description straight from the reference's own files, so that a C++ caller (include/BatchNeuralNet.h) needs
// nothing but the two paths cNeuralNet::LoadNet / LoadSolver get (learning/NeuralNet.cpp:62-79, 110-136; learning/NNSolver.cpp): the net prototxt (deploy
// or train form) gives the topology of the family (slice -> 3 x Convolution -> terr_ip0 -> trunk InnerProduct -> heads), the train prototxt the MemoryData
// batch_size and the per-blob lr_mult / decay_mult (`param { }` blocks, Caffe defaults 1 / 1), the solver prototxt base_lr / momentum / weight_decay /
// lr_policy and -- in the reference's own files -- the path of the train net (`net: "..."`).
// Plain text scanning (no protobuf here either): a layer's fields are looked up inside its `layer { ... }` block at any nesting depth, which covers both
// the reference's files (convolution_param { num_output: .. kernel_w: .. }) and the flattened digests under tests/golden/refdata.
#pragma once
#include <fstream>
#include <map>
#include <regex>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/dtrl_trainer.h"

namespace dtrl_tr {

struct TrainerFiles {
	dtrl_trainer_desc desc{};
	std::vector<float> lr_mult, decay_mult;   // per parameter element, Caffe blob order
	std::string train_net;                    // the file the multipliers and the batch size came from ("" = none found: Caffe defaults, batch 32)
};

namespace files_detail {
struct Layer { std::string name, type; int num_output = -1, kernel_w = -1, slice_point = -1, batch_size = -1, width = -1; std::vector<std::pair<float, float>> mults; };

inline bool ReadAll(const std::string& path, std::string& out)
{
	std::ifstream f(path);
	if (!f.is_open()) return false;
	std::stringstream ss; ss << f.rdbuf(); out = ss.str();
	out = std::regex_replace(out, std::regex("#[^\n]*"), "");
	return true;
}
inline std::vector<Layer> Layers(const std::string& txt)
{
	std::vector<Layer> layers;
	std::regex split("\\blayer\\s*\\{");
	std::sregex_token_iterator it(txt.begin(), txt.end(), split, -1), end;
	bool first = true;
	for (; it != end; ++it) {
		if (first) { first = false; continue; }
		const std::string blk = *it;
		Layer l; std::smatch m;
		if (std::regex_search(blk, m, std::regex("name:\\s*\"([^\"]+)\""))) l.name = m[1];
		if (std::regex_search(blk, m, std::regex("type:\\s*\"([^\"]+)\""))) l.type = m[1];
		auto geti = [&](const char* key, int& dst) { if (std::regex_search(blk, m, std::regex(std::string(key) + ":\\s*(\\d+)"))) dst = std::stoi(m[1]); };
		geti("num_output", l.num_output); geti("kernel_w", l.kernel_w); geti("slice_point", l.slice_point); geti("batch_size", l.batch_size); geti("width", l.width);
		std::regex pr("\\bparam\\s*\\{([^}]*)\\}");
		for (auto p = std::sregex_iterator(blk.begin(), blk.end(), pr); p != std::sregex_iterator(); ++p) {
			const std::string body = (*p)[1];
			float lr = 1.0f, dec = 1.0f; std::smatch q;
			if (std::regex_search(body, q, std::regex("lr_mult:\\s*([-0-9.eE]+)"))) lr = std::stof(q[1]);
			if (std::regex_search(body, q, std::regex("decay_mult:\\s*([-0-9.eE]+)"))) dec = std::stof(q[1]);
			l.mults.emplace_back(lr, dec);
		}
		layers.push_back(l);
	}
	return layers;
}
inline std::string Dir(const std::string& p) { const size_t k = p.find_last_of('/'); return k == std::string::npos ? std::string() : p.substr(0, k + 1); }
inline std::string ReplaceLast(std::string s, const std::string& a, const std::string& b)
{
	const size_t k = s.rfind(a);
	if (k == std::string::npos) return std::string();
	return s.replace(k, a.size(), b);
}
}  // namespace files_detail

// net_file: deploy or train prototxt; solver_file: solver prototxt ("" = no solver: evaluation only, batch 32, Caffe's solver defaults are not guessed).
// data_root: prefix for the `net:` path a solver names (the reference's files name it relative to the repository root); may be "".
inline bool ParseTrainerFiles(const std::string& net_file, const std::string& solver_file, const std::string& data_root, TrainerFiles& out, std::string& err)
{
	using namespace files_detail;
	std::string net_txt;
	if (!ReadAll(net_file, net_txt)) { err = "cannot open " + net_file; return false; }
	dtrl_trainer_desc& d = out.desc;
	d = dtrl_trainer_desc{};
	d.batch = 32; d.base_lr = 0.01f; d.momentum = 0.0f; d.weight_decay = 0.0f; d.discount = 0.9f;
	// solver
	std::string train_txt, train_path;
	if (!solver_file.empty()) {
		std::string s;
		if (!ReadAll(solver_file, s)) { err = "cannot open " + solver_file; return false; }
		std::smatch m;
		auto getf = [&](const char* key, float& dst) { if (std::regex_search(s, m, std::regex(std::string("(^|\\n)\\s*") + key + ":\\s*([-0-9.eE]+)"))) dst = std::stof(m[2]); };
		getf("base_lr", d.base_lr); getf("momentum", d.momentum); getf("weight_decay", d.weight_decay);
		if (std::regex_search(s, m, std::regex("lr_policy:\\s*\"([^\"]+)\"")) && m[1] != "fixed") { err = solver_file + ": lr_policy \"" + std::string(m[1]) + "\" (the native step implements \"fixed\", what the shipped solvers use)"; return false; }
		// the train net: the solver's `net:` entry, else <x>_solver -> <x>_train, else <net>_deploy -> <net>_train
		std::vector<std::string> cand;
		if (std::regex_search(s, m, std::regex("(^|\\n)\\s*net:\\s*\"([^\"]+)\""))) { cand.push_back(data_root.empty() ? std::string(m[2]) : data_root + "/" + std::string(m[2])); cand.push_back(m[2]); }
		cand.push_back(ReplaceLast(solver_file, "_solver", "_train"));
		cand.push_back(ReplaceLast(net_file, "_deploy", "_train"));
		for (const std::string& c : cand) if (!c.empty() && ReadAll(c, train_txt)) { train_path = c; break; }
	}
	std::vector<Layer> net = Layers(net_txt);
	std::vector<Layer> train = train_txt.empty() ? std::vector<Layer>() : Layers(train_txt);
	bool net_is_train = false;
	for (const Layer& l : net) if (l.type == "MemoryData") net_is_train = true;
	if (net_is_train && train.empty()) { train = net; train_path = net_file; }
	out.train_net = train_path;
	// input width
	int in_size = -1;
	{
		std::regex re("input_dim:\\s*(\\d+)");
		for (auto it = std::sregex_iterator(net_txt.begin(), net_txt.end(), re); it != std::sregex_iterator(); ++it) in_size = std::stoi((*it)[1]);
		for (const Layer& l : net) if (l.type == "MemoryData" && l.width > 0 && in_size < 0) in_size = l.width;
	}
	for (const Layer& l : train) if (l.type == "MemoryData") { if (l.batch_size > 0) { d.batch = l.batch_size; break; } }
	std::map<std::string, const Layer*> ips;
	std::vector<const Layer*> convs;
	d.n_terrain = -1;
	for (const Layer& l : net) {
		if (l.type == "Slice" && d.n_terrain < 0) d.n_terrain = l.slice_point;
		else if (l.type == "Convolution") convs.push_back(&l);
		else if (l.type == "InnerProduct") ips[l.name] = &l;
	}
	if (in_size <= 0 || d.n_terrain <= 0 || convs.size() != 3 || !ips.count("terr_ip0")) { err = net_file + ": not a net of the family (input, Slice, 3 x Convolution, terr_ip0, ...)"; return false; }
	d.state_size = in_size;
	for (int l = 0; l < 3; ++l) { d.conv_ch[l] = convs[l]->num_output; d.conv_k[l] = convs[l]->kernel_w; }
	d.fc_terr = ips["terr_ip0"]->num_output;
	std::vector<std::string> order = {"terr_conv0", "terr_conv1", "terr_conv2", "terr_ip0"};   // blob order (names as the family uses them; the convs are taken by position)
	if (ips.count("val_ip1") && ips.count("ip0") && ips.count("val_ip0") && ips.count("a0_ip1")) {
		d.fc_trunk = ips["ip0"]->num_output; d.fc_head = ips["val_ip0"]->num_output;
		d.n_frags = ips["val_ip1"]->num_output; d.frag_size = ips["a0_ip1"]->num_output;
		if (d.n_frags + 1 > 8) { err = "too many heads"; return false; }
		d.n_heads = 1 + d.n_frags; d.head_out[0] = d.n_frags;
		order.push_back("ip0"); order.push_back("val_ip0"); order.push_back("val_ip1");
		for (int f = 0; f < d.n_frags; ++f) { d.head_out[1 + f] = d.frag_size; order.push_back("a" + std::to_string(f) + "_ip0"); order.push_back("a" + std::to_string(f) + "_ip1"); }
	} else if (ips.count("ip1") && ips.count("ip2") && ips.count("output")) {
		d.fc_trunk = ips["ip1"]->num_output; d.fc_head = ips["ip2"]->num_output; d.n_heads = 1; d.n_frags = 0; d.frag_size = ips["output"]->num_output; d.head_out[0] = d.frag_size;
		order.push_back("ip1"); order.push_back("ip2"); order.push_back("output");
	} else { err = net_file + ": neither the MACE heads (ip0, val_ip0 / val_ip1, a<f>_ip0 / a<f>_ip1) nor the single head (ip1, ip2, output)"; return false; }
	for (size_t k = 3; k < order.size(); ++k) if (!ips.count(order[k])) { err = net_file + ": missing layer " + order[k]; return false; }   // (a net with fewer actor heads than val_ip1 outputs, a malformed prototxt: ADVICE r4)
	d.max_eval = 3 * d.batch;
	// per-element multipliers in blob order (weight blob, then bias blob, per layer)
	std::map<std::string, const Layer*> tl;
	std::vector<const Layer*> tconv;
	for (const Layer& l : train) { if (l.type == "Convolution") tconv.push_back(&l); else if (l.type == "InnerProduct") tl[l.name] = &l; }
	out.lr_mult.clear(); out.decay_mult.clear();
	int cin = 1, w = d.n_terrain;
	auto push = [&](const Layer* l, int64_t nw, int64_t nb) {
		const std::pair<float, float> mw = (l && l->mults.size() > 0) ? l->mults[0] : std::make_pair(1.0f, 1.0f);
		const std::pair<float, float> mb = (l && l->mults.size() > 1) ? l->mults[1] : std::make_pair(1.0f, 1.0f);
		out.lr_mult.insert(out.lr_mult.end(), static_cast<size_t>(nw), mw.first); out.decay_mult.insert(out.decay_mult.end(), static_cast<size_t>(nw), mw.second);
		out.lr_mult.insert(out.lr_mult.end(), static_cast<size_t>(nb), mb.first); out.decay_mult.insert(out.decay_mult.end(), static_cast<size_t>(nb), mb.second);
	};
	for (int l = 0; l < 3; ++l) {
		push(tconv.size() == 3 ? tconv[l] : nullptr, static_cast<int64_t>(d.conv_ch[l]) * cin * d.conv_k[l], d.conv_ch[l]);
		cin = d.conv_ch[l]; w = w - d.conv_k[l] + 1;
	}
	if (w <= 0) { err = "convolution kernels longer than the terrain slice"; return false; }
	int64_t nin = static_cast<int64_t>(cin) * w;
	for (size_t k = 3; k < order.size(); ++k) {
		const std::string& name = order[k];
		const int nout = ips[name]->num_output;
		int64_t fan_in;
		if (name == "terr_ip0") fan_in = nin;
		else if (name == "ip0" || name == "ip1") fan_in = d.fc_terr + (d.state_size - d.n_terrain);
		else if (name.size() > 4 && name.compare(name.size() - 4, 4, "_ip0") == 0) fan_in = d.fc_trunk;    // val_ip0, a<f>_ip0
		else if (name == "ip2") fan_in = d.fc_trunk;
		else fan_in = d.fc_head;                                                                          // val_ip1, a<f>_ip1, output
		push(tl.count(name) ? tl[name] : nullptr, static_cast<int64_t>(nout) * fan_in, nout);
	}
	return true;
}

}  // namespace dtrl_tr
