#pragma once
// forward declarations only (stand-in for third_party nlohmann_json, absent from this image)
namespace nlohmann {
template <typename T = void, typename SFINAE = void> struct adl_serializer;
class json;
using ordered_json = json;
}  // namespace nlohmann
