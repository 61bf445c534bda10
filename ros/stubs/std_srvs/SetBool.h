#ifndef STD_SRVS_SETBOOL_STUB_H
#define STD_SRVS_SETBOOL_STUB_H
#include <cstdint>
#include <string>
namespace std_srvs {
struct SetBoolRequest {
    uint8_t data = 0;
};
struct SetBoolResponse {
    uint8_t success = 0;
    std::string message;
};
struct SetBool {
    typedef SetBoolRequest Request;
    typedef SetBoolResponse Response;
};
}  // namespace std_srvs
#endif
