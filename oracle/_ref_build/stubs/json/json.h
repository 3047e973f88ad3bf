// Stand-in for jsoncpp's <json/json.h> -- TEST INFRASTRUCTURE (oracle/_ref_build), never part of the product.
// Provides the part of the Json::Value / Json::Reader API that the reference sources compiled by oracle/_ref_build/Makefile use
// (read-only access to parsed documents). Written from the public jsoncpp API; contains no jsoncpp source.
#pragma once
#include <cstdlib>
#include <istream>
#include <iterator>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace Json {

typedef unsigned int ArrayIndex;
typedef unsigned int UInt;
typedef int Int;
enum ValueType { nullValue = 0, intValue, uintValue, realValue, stringValue, booleanValue, arrayValue, objectValue };

class Value {
public:
	typedef std::vector<std::string> Members;
	Value() : type_(nullValue), num_(0), b_(false) {}
	Value(ValueType t) : type_(t), num_(0), b_(false) {}
	Value(int v) : type_(intValue), num_(v), b_(false) {}
	Value(unsigned v) : type_(uintValue), num_(v), b_(false) {}
	Value(double v) : type_(realValue), num_(v), b_(false) {}
	Value(bool v) : type_(booleanValue), num_(v ? 1 : 0), b_(v) {}
	Value(const char* s) : type_(stringValue), num_(0), b_(false), str_(s) {}
	Value(const std::string& s) : type_(stringValue), num_(0), b_(false), str_(s) {}

	ValueType type() const { return type_; }
	bool isNull() const { return type_ == nullValue; }
	bool isBool() const { return type_ == booleanValue; }
	bool isInt() const { return type_ == intValue; }
	bool isDouble() const { return type_ == realValue || type_ == intValue || type_ == uintValue; }
	bool isNumeric() const { return isDouble() || isBool(); }
	bool isString() const { return type_ == stringValue; }
	bool isArray() const { return type_ == arrayValue || type_ == nullValue; }
	bool isObject() const { return type_ == objectValue || type_ == nullValue; }
	bool empty() const { return size() == 0; }
	bool operator!() const { return isNull(); }
	ArrayIndex size() const { return type_ == arrayValue ? static_cast<ArrayIndex>(arr_.size()) : (type_ == objectValue ? static_cast<ArrayIndex>(obj_.size()) : 0); }

	double asDouble() const { return type_ == booleanValue ? (b_ ? 1.0 : 0.0) : num_; }
	float asFloat() const { return static_cast<float>(asDouble()); }
	int asInt() const { return static_cast<int>(asDouble()); }
	unsigned asUInt() const { return static_cast<unsigned>(asDouble()); }
	bool asBool() const { return type_ == booleanValue ? b_ : (type_ == nullValue ? false : num_ != 0); }
	std::string asString() const
	{
		if (type_ == stringValue) return str_;
		if (type_ == booleanValue) return b_ ? "true" : "false";
		if (type_ == nullValue) return "";
		std::ostringstream os; os << num_; return os.str();
	}

	const Value& operator[](ArrayIndex i) const { return (type_ == arrayValue && i < arr_.size()) ? arr_[i] : null(); }
	const Value& operator[](int i) const { return (*this)[static_cast<ArrayIndex>(i)]; }
	const Value& operator[](const char* key) const { return (*this)[std::string(key)]; }
	const Value& operator[](const std::string& key) const
	{
		if (type_ != objectValue) return null();
		for (size_t k = 0; k < keys_.size(); ++k) if (keys_[k] == key) return obj_[k];
		return null();
	}
	Value get(ArrayIndex i, const Value& dflt) const { const Value& v = (*this)[i]; return (&v == &null()) ? dflt : v; }
	Value get(int i, const Value& dflt) const { return get(static_cast<ArrayIndex>(i), dflt); }
	Value get(const char* key, const Value& dflt) const { return get(std::string(key), dflt); }
	Value get(const std::string& key, const Value& dflt) const { const Value& v = (*this)[key]; return (&v == &null()) ? dflt : v; }
	bool isMember(const std::string& key) const { return &(*this)[key] != &null(); }
	Members getMemberNames() const { return keys_; }

	// builder interface for the parser
	void append(const Value& v) { type_ = arrayValue; arr_.push_back(v); }
	void set(const std::string& key, const Value& v) { type_ = objectValue; keys_.push_back(key); obj_.push_back(v); }
private:
	static const Value& null() { static const Value n; return n; }
	ValueType type_; double num_; bool b_; std::string str_;
	std::vector<Value> arr_;
	std::vector<std::string> keys_; std::vector<Value> obj_;
};

class Reader {
public:
	bool parse(std::istream& is, Value& root, bool = true)
	{
		std::string text((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
		return parse(text, root);
	}
	bool parse(const std::string& text, Value& root, bool = true)
	{
		s_ = &text; p_ = 0; err_.clear();
		root = Value();
		if (!value(root)) return false;
		ws();
		return true;
	}
	std::string getFormattedErrorMessages() const { return err_; }
	std::string getFormatedErrorMessages() const { return err_; }
private:
	const std::string* s_ = nullptr; size_t p_ = 0; std::string err_;
	bool fail(const char* m) { if (err_.empty()) err_ = std::string(m) + " at offset " + std::to_string(p_); return false; }
	char cur() const { return p_ < s_->size() ? (*s_)[p_] : '\0'; }
	void ws()
	{
		for (;;) {
			while (p_ < s_->size() && (cur() == ' ' || cur() == '\t' || cur() == '\n' || cur() == '\r')) ++p_;
			if (cur() == '/' && p_ + 1 < s_->size() && (*s_)[p_ + 1] == '/') { while (p_ < s_->size() && cur() != '\n') ++p_; continue; }
			if (cur() == '/' && p_ + 1 < s_->size() && (*s_)[p_ + 1] == '*') { p_ += 2; while (p_ + 1 < s_->size() && !(cur() == '*' && (*s_)[p_ + 1] == '/')) ++p_; p_ += 2; continue; }
			break;
		}
	}
	bool str(std::string& out)
	{
		++p_; out.clear();
		while (p_ < s_->size() && cur() != '"') {
			if (cur() == '\\' && p_ + 1 < s_->size()) { const char e = (*s_)[p_ + 1]; out += (e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e); p_ += 2; }
			else out += (*s_)[p_++];
		}
		if (p_ >= s_->size()) return fail("unterminated string");
		++p_; return true;
	}
	bool value(Value& out)
	{
		ws();
		if (p_ >= s_->size()) return fail("unexpected end");
		const char c = cur();
		if (c == '{') {
			out = Value(objectValue); ++p_; ws();
			if (cur() == '}') { ++p_; return true; }
			for (;;) {
				ws(); std::string key;
				if (cur() != '"' || !str(key)) return fail("expected key");
				ws(); if (cur() != ':') return fail("expected ':'");
				++p_; Value v; if (!value(v)) return false;
				out.set(key, v);
				ws(); if (cur() == ',') { ++p_; continue; }
				if (cur() == '}') { ++p_; return true; }
				return fail("expected ',' or '}'");
			}
		}
		if (c == '[') {
			out = Value(arrayValue); ++p_; ws();
			if (cur() == ']') { ++p_; return true; }
			for (;;) {
				Value v; if (!value(v)) return false;
				out.append(v);
				ws(); if (cur() == ',') { ++p_; continue; }
				if (cur() == ']') { ++p_; return true; }
				return fail("expected ',' or ']'");
			}
		}
		if (c == '"') { std::string s; if (!str(s)) return false; out = Value(s); return true; }
		if (s_->compare(p_, 4, "true") == 0) { out = Value(true); p_ += 4; return true; }
		if (s_->compare(p_, 5, "false") == 0) { out = Value(false); p_ += 5; return true; }
		if (s_->compare(p_, 4, "null") == 0) { out = Value(); p_ += 4; return true; }
		char* end = nullptr;
		const double v = std::strtod(s_->c_str() + p_, &end);
		if (end == s_->c_str() + p_) return fail("bad value");
		bool is_int = true;
		for (const char* q = s_->c_str() + p_; q < end; ++q) if (*q == '.' || *q == 'e' || *q == 'E') is_int = false;
		out = is_int ? Value(static_cast<int>(v)) : Value(v);
		if (is_int && static_cast<double>(static_cast<int>(v)) != v) out = Value(v);
		p_ = static_cast<size_t>(end - s_->c_str());
		return true;
	}
};

}  // namespace Json
