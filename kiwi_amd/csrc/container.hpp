// Section container reader/writer (python twin: kiwi_amd/container.py).
// Layout: 8-byte kind tag, u32 nSections, u32 reserved, nSections x {char name[32]; u64 offset; u64 nbytes},
// payloads 64-byte aligned.  Used for the raw model file and for flat-model dumps.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace kamd
{
	struct Section { const uint8_t* data = nullptr; size_t size = 0; };

	class Container
	{
		std::vector<uint8_t> buf;
		std::map<std::string, Section> secs;
		char kindTag[9] = { 0 };
	public:
		void load(const std::string& path)
		{
			FILE* f = std::fopen(path.c_str(), "rb");
			if (!f) throw std::runtime_error{ "cannot open model file: " + path };
			std::fseek(f, 0, SEEK_END);
			long n = std::ftell(f);
			std::fseek(f, 0, SEEK_SET);
			buf.resize((size_t)n);
			if (n && std::fread(buf.data(), 1, (size_t)n, f) != (size_t)n) { std::fclose(f); throw std::runtime_error{ "short read: " + path }; }
			std::fclose(f);
			parse();
		}

		void parse()
		{
			if (buf.size() < 16) throw std::runtime_error{ "model file truncated" };
			std::memcpy(kindTag, buf.data(), 8);
			uint32_t n;
			std::memcpy(&n, buf.data() + 8, 4);
			if (16 + (size_t)n * 48 > buf.size()) throw std::runtime_error{ "model file: bad section table" };
			secs.clear();
			for (uint32_t i = 0; i < n; ++i)
			{
				const uint8_t* e = buf.data() + 16 + (size_t)i * 48;
				char name[33] = { 0 };
				std::memcpy(name, e, 32);
				uint64_t off, nb;
				std::memcpy(&off, e + 32, 8);
				std::memcpy(&nb, e + 40, 8);
				if (off + nb > buf.size()) throw std::runtime_error{ std::string{ "model file: section out of range: " } + name };
				secs[name] = Section{ buf.data() + off, (size_t)nb };
			}
		}

		const char* kind() const { return kindTag; }
		bool has(const std::string& name) const { return secs.count(name) != 0; }
		Section get(const std::string& name) const
		{
			auto it = secs.find(name);
			if (it == secs.end()) throw std::runtime_error{ "model file: missing section " + name };
			return it->second;
		}
		template<class T> const T* ptr(const std::string& name, size_t* count = nullptr) const
		{
			auto s = get(name);
			if (count) *count = s.size / sizeof(T);
			return reinterpret_cast<const T*>(s.data);
		}
	};

	class ContainerWriter
	{
		struct Item { std::string name; std::vector<uint8_t> bytes; };
		std::vector<Item> items;
	public:
		void add(const std::string& name, const void* data, size_t nbytes)
		{
			Item it; it.name = name; it.bytes.assign((const uint8_t*)data, (const uint8_t*)data + nbytes);
			items.emplace_back(std::move(it));
		}
		template<class T> void add(const std::string& name, const std::vector<T>& v) { add(name, v.data(), v.size() * sizeof(T)); }
		void save(const std::string& path, const char kind[8]) const
		{
			FILE* f = std::fopen(path.c_str(), "wb");
			if (!f) throw std::runtime_error{ "cannot write: " + path };
			uint32_t n = (uint32_t)items.size(), zero = 0;
			std::fwrite(kind, 1, 8, f); std::fwrite(&n, 4, 1, f); std::fwrite(&zero, 4, 1, f);
			uint64_t off = 16 + 48ull * n;
			std::vector<uint64_t> offs;
			for (auto& it : items)
			{
				off = (off + 63) & ~63ull;
				offs.push_back(off);
				char name[32] = { 0 };
				std::strncpy(name, it.name.c_str(), 31);
				uint64_t nb = it.bytes.size();
				std::fwrite(name, 1, 32, f); std::fwrite(&off, 8, 1, f); std::fwrite(&nb, 8, 1, f);
				off += nb;
			}
			for (size_t i = 0; i < items.size(); ++i)
			{
				std::fseek(f, (long)offs[i], SEEK_SET);
				if (!items[i].bytes.empty()) std::fwrite(items[i].bytes.data(), 1, items[i].bytes.size(), f);
			}
			std::fclose(f);
		}
	};
}
