#ifndef DYNAMIC_RECONFIGURE_SERVER_STUB_H
#define DYNAMIC_RECONFIGURE_SERVER_STUB_H
#include <cstdint>
#include <functional>
namespace dynamic_reconfigure {
template <class ConfigType>
class Server {
   public:
    typedef std::function<void(ConfigType &, uint32_t level)> CallbackType;  // (boost::function in roscpp)
    void setCallback(const CallbackType &callback);
};
}  // namespace dynamic_reconfigure
#endif
